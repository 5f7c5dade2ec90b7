"""Pins the oracle's helpers against every known-answer test the reference holds for this path
(SURVEY.md section 8c): src/types.rs:454-488 and src/meancov_estimation.rs:450-533.  These are the
only golden vectors the reference owns; everything downstream of the tree walk is unpinned."""
import ctypes as C

import numpy as np


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


def mean_cov64(lib, pts):
    pts = _d(pts)
    mean, cov = np.zeros(3), np.zeros(9)
    assert lib.orc_mean_cov_f64(_ptr(pts), C.c_uint32(len(pts)), _ptr(mean), _ptr(cov)) == 1
    return mean, cov.reshape(3, 3)


def test_intrinsic(oracle):
    """src/types.rs:476-488 test_intrinsic"""
    lib = oracle.lib()
    K = _f([[22.0, 11.4, 12.11], [2.1, 4.1, 2.11], [1.3, 3.1, 19.0]])
    p3 = _f([11.0, 12.0, 32.2])
    p2 = np.zeros(2, dtype=np.float32)
    lib.orc_space_to_img(_ptr(K), _ptr(p3), _ptr(p2))
    assert abs(p2[0] - 1.15896578) < 1e-4 and abs(p2[1] - 0.21143073) < 1e-4
    back = np.zeros(3, dtype=np.float32)
    lib.orc_img_to_space(_ptr(K), _ptr(p2), C.c_float(p3[2]), _ptr(back))
    assert np.all(np.abs(back - p3) < 1e-4)


def test_mean_cov3(oracle):
    """src/meancov_estimation.rs:461-490 test_mean_cov3 (both sets, incl. the determinant)"""
    lib = oracle.lib()
    m, c = mean_cov64(lib, [[1.0, 2.0, 3.0], [1.2, 1.0, 3.2], [-1.0, -2.1, 3.0], [0.0, 1.0, 0.0]])
    assert np.allclose(m, [0.3, 0.475, 2.3], atol=1e-3)
    exp = np.array([[1.0266666, 1.576666, 0.36], [1.576666, 3.1691666, -0.49], [0.36, -0.49, 2.36]])
    assert np.all(np.abs(c - exp) < 1e-3)
    pts = [[-32.48225021362305, 24.72743034362793, -3.9425208568573],
           [-25.82341957092285, -25.307233810424805, 1.955498456954956],
           [35.37421417236328, -18.529083251953125, -5.888242721557617],
           [43.30265808105469, -60.69481658935547, -15.176074028015137],
           [32.97354507446289, -7.171285629272461, -3.897606134414673]]
    _, c = mean_cov64(lib, pts)
    exp = np.array([[1341.63076476, -685.47821414, -157.223241746],
                    [-685.478214144, 954.396794746, 110.60252659],
                    [-157.223241746, 110.60252659, 38.573568346]])
    assert np.all(np.abs(c - exp) < 1e-3)
    lib.orc_mat3_det_f64.restype = C.c_double
    assert abs(lib.orc_mat3_det_f64(_ptr(_d(c))) - 15102509.494226849) < 1e-4


def test_mean_cov2_embedded(oracle):
    """src/meancov_estimation.rs:450-460 test_mean_cov2: the 2-D set embedded in 3-D (z = 0); the
    generic estimator is the same code for every dimension."""
    lib = oracle.lib()
    m, c = mean_cov64(lib, [[2, 6, 0], [3, 4, 0], [3, 8, 0], [4, 6, 0]])
    assert np.allclose(m[:2], [3, 6], atol=1e-3)
    assert abs(c[0, 0] - 0.66666) < 1e-3 and abs(c[1, 1] - 2.6666) < 1e-3
    assert abs(c[0, 1]) < 1e-3 and abs(c[1, 0]) < 1e-3


def test_det_trace(oracle):
    """src/meancov_estimation.rs:492-500 test_det_2_3_trace (3x3 part)"""
    lib = oracle.lib()
    m3 = _d([[1, 3, 22], [2, 44, 1], [2, 0, 3.1]])
    assert abs(lib.orc_mat3_det_f64(_ptr(m3)) - (-1812.199)) < 1e-3
    assert abs(lib.orc_trace_f64(_ptr(m3)) - 48.1) < 1e-3


def test_inverse(oracle):
    """src/meancov_estimation.rs:502-515 test_inverse (3x3 part), f64 and the f32 instantiation"""
    lib = oracle.lib()
    m3 = _d([[2.3, 1.4, 12.11], [2.1, 44.11, 2.11], [1.3, 4.1, 19.0]])
    exp = np.array([[0.65540671, 0.01821446, -0.4197583], [-0.02936075, 0.02209108, 0.01626034],
                    [-0.03850788, -0.00601328, 0.07784307]])
    out = np.zeros(9)
    lib.orc_mat3_inv_f64(_ptr(m3), _ptr(out))
    assert np.all(np.abs(out.reshape(3, 3) - exp) < 1e-3)
    out32 = np.zeros(9, dtype=np.float32)
    lib.orc_mat3_inv_f32(_ptr(_f(m3)), _ptr(out32))
    assert np.all(np.abs(out32.reshape(3, 3) - exp) < 1e-3)


def test_mat_vec_mul(oracle):
    """src/meancov_estimation.rs:517-525 test_mat_vec_mul (3x3 part)"""
    lib = oracle.lib()
    m3 = _d([[1.3, 12.1, 2.3], [3.1, 33.1, 14.1], [1.0, 2.0, 3.0]])
    out = np.zeros(3)
    lib.orc_mat3_vec_f64(_ptr(m3), _ptr(_d([11.0, 12.0, 32.2])), _ptr(out))
    assert np.all(np.abs(out - [233.56, 885.32, 131.6]) < 1e-3)


def test_transposed_matrix(oracle):
    """src/meancov_estimation.rs:527-533 test_transposed_matrix (3-D part): v * v^T"""
    lib = oracle.lib()
    out = np.zeros(9)
    lib.orc_outer_f64(_ptr(_d([2.0, 1.1, 4.3])), _ptr(out))
    assert np.all(np.abs(out.reshape(3, 3) - [[4.0, 2.2, 8.6], [2.2, 1.21, 4.73], [8.6, 4.73, 18.49]]) < 1e-3)


def test_rect_accessors_and_average(oracle):
    """src/types.rs:454-474 test_rect (the accessors the hot path uses: x, y, width, height, size)
    expressed on the flat x0,y0,x1,y1 form, plus average_value_in_rect (types.rs:317-339) on a
    ramp whose mean is known in closed form."""
    lib = oracle.lib()
    # Rect::new(1, 2, 10, 20) -> topleft (1,2), bottomright (11,22)
    r = np.array([1, 2, 1 + 10, 2 + 20], dtype=np.uint16)
    assert (r[2] - r[0], r[3] - r[1]) == (10, 20) and int(r[2] - r[0]) * int(r[3] - r[1]) == 200
    img = (np.arange(40 * 30, dtype=np.uint32).reshape(30, 40) % 65536).astype(np.uint16)
    got = lib.orc_average_value_in_rect(_ptr(img), C.c_uint32(40), C.c_uint32(3), C.c_uint32(4), _ptr(r))
    exp = img[4 + 2:4 + 22, 3 + 1:3 + 11].astype(np.float64).mean()
    assert got == exp
    empty = np.array([5, 5, 5, 9], dtype=np.uint16)   # zero width -> count == 0 -> 0.0 (types.rs:335-337)
    assert lib.orc_average_value_in_rect(_ptr(img), C.c_uint32(40), C.c_uint32(0), C.c_uint32(0), _ptr(empty)) == 0.0


def test_kernel_table(oracle):
    """FullArray3D::build_kernel (src/meanshift.rs:228-252): centre is exp(0) = 1, symmetric, and
    `gaussian_sigma` enters as the variance (prediction.rs:314)."""
    lib = oracle.lib()
    k = np.zeros(8000, dtype=np.float32)
    lib.orc_build_kernel(C.c_uint32(20), C.c_float(8.0), _ptr(k))
    k3 = k.reshape(20, 20, 20)   # [z][y][x]
    assert k3[10, 10, 10] == 1.0
    assert k3[10, 10, 11] == np.float32(np.exp(np.float32(-1.0) / np.float32(16.0)))
    assert np.array_equal(k3, k3.transpose(2, 1, 0)) and np.array_equal(k3[1:], k3[1:][::-1])
    assert k3[0, 0, 0] == np.float32(np.exp(np.float32(-300.0) / np.float32(16.0)))
