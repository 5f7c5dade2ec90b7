"""SURVEY 8f row N4, second half: HoughPrediction::build_hough_image in full (prediction.rs:760-845: votes + imageproc's
gaussian_blur_f32) and predict_parameter_from2dhough (:343-367: last-max argmax, img_to_space_coord).

PARITY UNPINNED: imageproc 0.12.0 is an external crate whose source is not in the container (Cargo.lock:555); the blur
restates its published algorithm (oracle/dh_oracle.c).  The CPU tests check the oracle's blur against an independent
numpy statement of the same definition and against properties the definition implies; the GPU tests compare the HIP
path with the oracle: the blurred image is a u16 image, so "within 1e-4" means equal, the argmax is exact."""
import numpy as np
import pytest

from depthhead_amd import synth


def _numpy_blur(img, sigma):
    """The same definition, vectorised differently: f32 taps, per pass acc += px * k in tap order, clamp + truncate."""
    radius = int(np.ceil(np.float32(2.0) * np.float32(sigma)))
    x = np.arange(radius + 1, dtype=np.float32)
    s = np.float32(sigma)
    half = (np.float32(1.0) / (np.sqrt(np.float32(2.0) * np.float32(np.pi)) * s)) * np.exp(-(x * x) / (np.float32(2.0) * (s * s)))
    k = np.concatenate([half[:0:-1], half]).astype(np.float32)

    def one_pass(a, axis):
        n = a.shape[axis]
        acc = np.zeros(a.shape, dtype=np.float32)
        for i, kv in enumerate(k):
            idx = np.clip(np.arange(n) + i - len(k) // 2, 0, n - 1)
            acc = (acc + np.take(a, idx, axis=axis).astype(np.float32) * kv).astype(np.float32)
        out = np.where(acc < 65535.0, np.where(acc > 0.0, np.trunc(acc), 0.0), 65535.0)
        return out.astype(np.uint16)

    return one_pass(one_pass(img, 1), 0), k


@pytest.mark.parametrize("sigma", [8.0, 0.5, 2.25, 1e-3])
def test_oracle_blur_against_numpy_restatement(oracle, sigma):
    rs = np.random.RandomState(7)
    img = (rs.rand(37, 53) < 0.08) * rs.randint(1, 65536, (37, 53))
    img[0, 0] = 65535; img[-1, -1] = 65535; img[5, :] = 65535              # saturating sums at sigma < 1
    img = img.astype(np.uint16)
    want, k = _numpy_blur(img, sigma)
    kk = oracle.gaussian_kernel(sigma)
    assert kk.size == 2 * int(np.ceil(2 * sigma)) + 1 and np.array_equal(kk, kk[::-1])
    # numpy's exp / sqrt may differ from libm's by an ulp: taps within 2 ulp, images equal up to that (+-1 count)
    assert np.allclose(kk, k, rtol=3e-7, atol=0)
    got = oracle.gaussian_blur_u16(img, sigma)
    assert np.abs(got.astype(np.int64) - want.astype(np.int64)).max() <= 1


def test_blur_properties(oracle):
    """What the definition implies: a constant image keeps its value times the (unnormalised) tap sum squared, borders
    replicate, the result is symmetric for a symmetric input, sigma <= 0 is refused (the crate asserts)."""
    k = oracle.gaussian_kernel(3.0)
    s = np.float32(0)
    for v in k:
        s = np.float32(s + np.float32(1000.0) * v)
    const = np.full((20, 30), 1000, dtype=np.uint16)
    out = oracle.gaussian_blur_u16(const, 3.0)
    row = np.uint16(np.trunc(s))
    s2 = np.float32(0)
    for v in k:
        s2 = np.float32(s2 + np.float32(row) * v)
    assert np.all(out == np.uint16(np.trunc(s2)))
    img = np.zeros((31, 31), dtype=np.uint16)
    img[15, 15] = 50000
    b = oracle.gaussian_blur_u16(img, 2.0)
    assert np.array_equal(b, b[::-1]) and np.array_equal(b, b[:, ::-1]) and np.array_equal(b, b.T) and b[15, 15] == b.max()
    with pytest.raises(ValueError):
        oracle.gaussian_blur_u16(img, 0.0)


def test_last_max_wins(oracle):
    """max_by_key keeps the LAST of equal maxima (prediction.rs:351-356): a frame without any vote gives an all-zero image,
    so the argmax is the bottom-right pixel."""
    forest = synth.synth_forest(3, 4, synth.FOREST_SEED_BASE + 31)
    model = synth.ModelParams(stepwidth=8)
    img = np.zeros((120, 160), dtype=np.uint16)
    img[119, 159] = 1234                                                   # the depth read at the argmax pixel
    K = synth.default_intrinsic(160, 120)
    mid, rot = oracle.predict_from2dhough(forest, model, img, K)
    Kinv = np.linalg.inv(K.astype(np.float64))
    want = (Kinv @ np.array([159.0, 119.0, 1.0])) * 1234.0
    assert np.allclose(mid, want, rtol=1e-5) and np.all(rot == 0)


# ------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def hp_mod(hip_lib):
    from depthhead_amd import prediction
    return prediction


@pytest.mark.gpu
@pytest.mark.parametrize("general", [False, True])
@pytest.mark.parametrize("w,h,step,sigma", [(320, 240, 4, 8.0), (200, 160, 7, 2.5), (168, 128, 1, 0.7), (240, 200, 10, 12.0)])
def test_blurred_hough_image_and_2d_prediction(hp_mod, oracle, w, h, step, sigma, general):
    import os
    forest = synth.fit_forest(6, 10, synth.FOREST_SEED_BASE + 170, n_frames=12, subset=1500)
    assert (forest.leaf_prob >= 0.95).any()
    model = synth.ModelParams(stepwidth=step, gaussian_sigma=sigma)
    frames = synth.biwi_batch(4, w, h, first=80)
    frames[2, : h // 2] = 0                                               # half the frame background
    frames[3] = 0                                                         # no vote at all: all-zero image, last pixel wins
    frames[3, h - 1, w - 1] = 900
    K = synth.default_intrinsic(w, h)
    if general:
        os.environ["DH_FORCE_GENERAL"] = "1"
    try:
        with hp_mod.HoughPrediction(forest, model, device=0) as hp:
            blurred = hp.build_hough_image(frames, hp_mod.IntrinsicMatrix(K))
            poses = hp.predict_parameter_from2dhough(frames, hp_mod.IntrinsicMatrix(K))
            one = hp.predict_parameter_from2dhough(frames[0], hp_mod.IntrinsicMatrix(K))
            votes = hp.build_hough_votes(frames, hp_mod.IntrinsicMatrix(K))
            p3 = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))       # the main path still works afterwards
    finally:
        os.environ.pop("DH_FORCE_GENERAL", None)
    assert blurred.max() > 0 and votes.max() > 0
    for i in range(4):
        ref_img = oracle.build_hough_image(forest, model, frames[i], K)
        assert np.max(np.abs(blurred[i].astype(np.float64) - ref_img.astype(np.float64))) <= 1e-4, (w, h, i)
        assert np.array_equal(blurred[i], ref_img)
        assert np.array_equal(blurred[i], oracle.gaussian_blur_u16(votes[i], sigma))
        mid, rot = oracle.predict_from2dhough(forest, model, frames[i], K)
        assert np.array_equal(poses["mid_point"][i], mid), (i, poses["mid_point"][i], mid)      # same argmax pixel, same f32 arithmetic
        assert np.max(np.abs(poses["mid_point"][i] - mid)) <= 1e-4 and np.all(poses["rotation"][i] == 0.0) and np.all(rot == 0.0)
    assert np.array_equal(one.mid_point, poses["mid_point"][0]) and one.bounding_box == (0, 0, 0, 0)
    ref = oracle.predict_batch(forest, model, frames, K)
    assert np.array_equal(p3["mid_point"], ref["mid_point"]) and np.array_equal(p3["rotation"], ref["rotation"])


@pytest.mark.gpu
def test_2d_variant_needs_positive_sigma(hp_mod):
    from depthhead_amd._lib import DepthheadError
    forest = synth.synth_forest(3, 4, synth.FOREST_SEED_BASE + 31)
    frames = synth.biwi_batch(1, 160, 120)
    with hp_mod.HoughPrediction(forest, synth.ModelParams(stepwidth=8, gaussian_sigma=0.0), device=0) as hp:
        with pytest.raises(DepthheadError) as ei:
            hp.predict_parameter_from2dhough(frames, hp_mod.IntrinsicMatrix(synth.default_intrinsic(160, 120)))
        assert ei.value.code == -1
        hp.predict_batch(frames, hp_mod.IntrinsicMatrix(synth.default_intrinsic(160, 120)))   # the 3-D path accepts sigma 0 (expf(-inf) taps)
