"""The C-ABI library without a GPU: it loads, exports every symbol include/depthhead_hip.h declares,
its POD layouts match the header, and the host-only entry points (forest validation, patch grid)
behave.  No compute call is made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from depthhead_amd import synth, _lib
from depthhead_amd.forest import Forest, NODE_DTYPE

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "depthhead_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dh_[a-z_0-9]+)\s*\(", src)))


def test_exports_every_declared_symbol(hip_lib):
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in depthhead_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == names, "depthhead_amd/_lib.py EXPORTS out of sync with the header"
    assert hip_lib.dh_version() == 100


def test_pod_layouts_match_header():
    assert NODE_DTYPE.itemsize == 32 and NODE_DTYPE.fields["threshold"][1] == 16 and NODE_DTYPE.fields["child_one"][1] == 28
    assert _lib.POSE_DTYPE.itemsize == 40 and _lib.POSE_DTYPE.fields["rotation"][1] == 16
    assert C.sizeof(_lib.Params) == 20 and C.sizeof(_lib.Timing) == 32
    assert C.sizeof(_lib.ForestDesc) == 80


def _desc(f):
    return _lib.ForestDesc(f.n_trees, f.roots.ctypes.data, f.n_nodes, f.nodes.ctypes.data, f.n_leaves,
                           f.leaf_prob.ctypes.data, f.off_begin.ctypes.data, f.rot_begin.ctypes.data,
                           f.offsets.ctypes.data, f.rotations.ctypes.data)


def _create(lib, f):
    h = C.c_void_p()
    rc = lib.dh_forest_create(C.byref(_desc(f)), C.byref(h))
    if rc == 0:
        lib.dh_forest_destroy(h)
    return rc, lib.dh_last_error().decode()


def _clone(f):
    return Forest(f.roots.copy(), f.nodes.copy(), f.leaf_prob.copy(), f.off_begin.copy(), f.rot_begin.copy(),
                  f.offsets.copy(), f.rotations.copy())


def test_forest_validation(hip_lib):
    """dh_forest_create rejects exactly the inputs on which the reference would panic or read out
    of bounds (header comment), and reports why."""
    good = synth.synth_forest(3, 5, 7)
    assert _create(hip_lib, good)[0] == 0
    info = [C.c_uint32() for _ in range(4)]
    h = C.c_void_p()
    assert hip_lib.dh_forest_create(C.byref(_desc(good)), C.byref(h)) == 0
    assert hip_lib.dh_forest_info(h, *[C.byref(i) for i in info]) == 0
    assert (info[0].value, info[1].value, info[2].value) == (3, good.n_nodes, good.n_leaves)
    assert info[3].value == good.max_depth() == 5
    hip_lib.dh_forest_destroy(h)

    f = _clone(good); f.nodes["child_one"][0] = good.n_nodes + 5
    rc, msg = _create(hip_lib, f); assert rc == -2 and "child out of range" in msg
    f = _clone(good); f.nodes["child_zero"][0] = 0            # cycle: root points at itself
    rc, msg = _create(hip_lib, f); assert rc == -2 and "twice" in msg
    f = _clone(good); f.nodes["child_zero"][1] = f.nodes["child_one"][1]   # shared subtree
    rc, msg = _create(hip_lib, f); assert rc == -2
    f = _clone(good); f.nodes["r1"][2] = (30, 10, 20, 40)     # x1 < x0: u32 underflow in Rect::width (types.rs:47)
    rc, msg = _create(hip_lib, f); assert rc == -2 and "negative extent" in msg
    f = _clone(good); f.roots[1] = ~(good.n_leaves + 1)
    rc, msg = _create(hip_lib, f); assert rc == -2 and "root" in msg
    f = _clone(good); f.off_begin[3] = f.off_begin[4] + 1
    rc, msg = _create(hip_lib, f); assert rc == -2 and "monotone" in msg
    # a positive leaf without votes: the reference divides by offsets.len() == 0 (prediction.rs:594)
    L = int(np.flatnonzero(good.leaf_prob == 0)[0])
    f = _clone(good); f.leaf_prob[L] = 0.5
    rc, msg = _create(hip_lib, f); assert rc == -2 and "no offsets" in msg
    # a rotation whose bin leaves [0,120) after one wrap: index out of bounds at prediction.rs:636
    f = _clone(good); f.rotations[0] = (1000.0, 0.0, 0.0)
    rc, msg = _create(hip_lib, f); assert rc == -2 and "rotation bin" in msg
    f = _clone(good); f.rotations[0] = (539.0, -542.9, 0.0)   # still inside after the single wrap
    assert _create(hip_lib, f)[0] == 0
    assert hip_lib.dh_forest_create(None, C.byref(h)) == -1


def test_forest_create_survives_random_corruption(hip_lib):
    """2000 randomly corrupted forest descriptions (child links, roots, rectangles, CSR offsets, probabilities,
    NaN / inf thresholds and votes): dh_forest_create answers every one with DH_OK or a clean DH_EFOREST /
    DH_EINVAL -- never a crash -- and everything it accepts is structurally safe to walk."""
    rs = np.random.RandomState(2026)
    good = synth.synth_forest(4, 6, 11)
    accepted = 0
    for it in range(2000):
        f = _clone(good)
        for _ in range(int(rs.randint(1, 4))):
            kind = int(rs.randint(0, 9))
            if kind == 0:
                f.nodes["child_zero"][rs.randint(f.n_nodes)] = int(rs.randint(-3 * f.n_leaves, 3 * f.n_nodes))
            elif kind == 1:
                f.nodes["child_one"][rs.randint(f.n_nodes)] = int(rs.randint(-3 * f.n_leaves, 3 * f.n_nodes))
            elif kind == 2:
                f.roots[rs.randint(f.n_trees)] = int(rs.randint(-2 * f.n_leaves, 2 * f.n_nodes))
            elif kind == 3:
                f.nodes["r1"][rs.randint(f.n_nodes)] = tuple(int(v) for v in rs.randint(0, 200, 4))
            elif kind == 4:
                f.nodes["r2"][rs.randint(f.n_nodes)] = tuple(int(v) for v in rs.randint(0, 65536, 4))
            elif kind == 5:
                f.off_begin[rs.randint(f.n_leaves)] = int(rs.randint(0, 2 * f.offsets.shape[0] + 2))     # (the last entry sizes the caller's array: left intact)
            elif kind == 6:
                f.rot_begin[rs.randint(f.n_leaves)] = int(rs.randint(0, 2 * f.rotations.shape[0] + 2))
            elif kind == 7:
                f.nodes["threshold"][rs.randint(f.n_nodes)] = rs.choice([np.nan, np.inf, -np.inf, 1e300, -1e300])
            else:
                f.leaf_prob[rs.randint(f.n_leaves)] = rs.choice([np.nan, -1.0, 2.0, 0.5, 0.0])
        rc, msg = _create(hip_lib, f)
        assert rc in (0, -1, -2), (it, rc, msg)
        if rc == 0:
            accepted += 1
            # accepted forests are trees: every walk from every root terminates within n_nodes steps
            for r in f.roots:
                seen, stack = 0, [int(r)]
                while stack:
                    c = stack.pop()
                    if c < 0:
                        assert ~c < f.n_leaves
                        continue
                    assert c < f.n_nodes
                    seen += 1
                    assert seen <= f.n_nodes
                    stack += [int(f.nodes["child_zero"][c]), int(f.nodes["child_one"][c])]
        else:
            assert msg
    assert accepted > 0


def test_patch_grid_matches_reference_loops(hip_lib):
    """dh_patch_grid against a literal transcription of the while loops (prediction.rs:535-548, 684-686)."""
    def loops(w, h, s, sw, sh):
        lw, lh = sw // 2, sh // 2
        rw, rh = sw - lw, sh - lh
        nx = ny = 0
        x = lw
        while x < w - rw:
            nx += 1; x += s
        y = lh
        while y < h - rh:
            ny += 1; y += s
        return nx, ny
    for (w, h, s, sw, sh) in [(640, 480, 10, 80, 80), (640, 480, 4, 80, 80), (640, 480, 2, 80, 80), (320, 240, 1, 80, 80),
                              (80, 80, 4, 80, 80), (81, 83, 1, 80, 80), (200, 100, 7, 61, 47), (97, 99, 3, 33, 32)]:
        prm = _lib.Params(s, sw, sh, 8.0, 20)
        nx, ny = C.c_int(), C.c_int()
        assert hip_lib.dh_patch_grid(C.byref(prm), w, h, C.byref(nx), C.byref(ny)) == 0
        assert (nx.value, ny.value) == loops(w, h, s, sw, sh) == synth.ModelParams(s, sw, sh).patch_grid(w, h)
    prm = _lib.Params(4, 80, 80, 8.0, 20)
    nx, ny = C.c_int(), C.c_int()
    assert hip_lib.dh_patch_grid(C.byref(prm), 79, 480, C.byref(nx), C.byref(ny)) == -5   # frame narrower than the patch
    assert "smaller than" in hip_lib.dh_last_error().decode()
    prm0 = _lib.Params(0, 80, 80, 8.0, 20)
    assert hip_lib.dh_patch_grid(C.byref(prm0), 640, 480, C.byref(nx), C.byref(ny)) == -1


def test_predictor_create_fails_loudly_without_gpu(hip_lib):
    """There is no CPU fallback: without a device the product path reports an error."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from depthhead_amd.prediction import HoughPrediction
    with pytest.raises(_lib.DepthheadError) as ei:
        HoughPrediction(synth.synth_forest(2, 4, 3), synth.ModelParams())
    assert ei.value.code in (-3, -1)


def test_product_code_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "depthhead_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "pyoracle" not in txt and "dh_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, fn


def test_product_library_has_no_profiling_knobs(hip_lib):
    """The kernel-truncating switches (DH_TRAV_STOP, ... -- "results invalid") are compiled only into the
    tools/ twin built with -DDH_PROFILING_KNOBS; the product library does not even contain their names, and
    the environment is read in exactly one function (dh_read_knobs_ in dh_host.cpp, called by dh_predictor_create)."""
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"DH_TRAV_STOP", b"DH_EMIT_STOP", b"DH_VOTE_STOP", b"DH_CL_STOP", b"DH_TRAV_STAMPS"):
        assert name not in blob, name
    assert b"DH_FORCE_GENERAL" in blob     # (the result-preserving diagnostic switches are still there)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "depthhead_amd", "csrc")
    host = open(os.path.join(csrc, "dh_host.cpp")).read()
    body = host[host.index("Knobs dh_read_knobs_()"):]
    body = body[:body.index("\n}\n") + 3]
    assert host.count("getenv(") == body.count("getenv(") > 0, "getenv outside dh_read_knobs_()"
    for fn in sorted(os.listdir(csrc)):
        if fn != "dh_host.cpp":
            assert "getenv(" not in open(os.path.join(csrc, fn)).read(), fn


def test_cpp_example_compiles_and_links(tmp_path):
    """examples/predict_frame.cpp -- a host with no Python in it -- builds against the header and links the
    shared library (it needs a GPU to run: tests/test_gpu_parity.py::test_cpp_example_runs)."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "depthhead_amd")
    if not os.path.exists(os.path.join(libdir, "libdepthhead_hip.so")):
        from depthhead_amd import build
        build.build()
    exe = str(tmp_path / "predict_frame")
    cmd = [gxx, "-std=c++17", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "predict_frame.cpp"),
           "-L" + libdir, "-ldepthhead_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert os.path.exists(exe)


def test_every_entry_point_runs_inside_the_exception_guard():
    """include/depthhead_hip.h: "never throws or aborts across the boundary".  Every `extern "C" int dh_*` definition of the
    library is either generated by the DH_API macro (dh_api.hip: body inside dh_guard_) or calls dh_guard_ itself (dh_biwi.cpp);
    the only unguarded definitions are the two that cannot throw (dh_version, dh_last_error)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "depthhead_amd", "csrc")
    api = open(os.path.join(csrc, "dh_api.hip")).read()
    guarded = set("dh_" + m for m in re.findall(r"^DH_API\((\w+),", api, flags=re.M))
    assert "dh_guard_(" in api[api.index("#define DH_API"):api.index("#define DH_API") + 300]
    raw = set(re.findall(r'extern "C" (?:int|const char \*)\s*(dh_\w+)\(', api))
    assert raw <= {"dh_version", "dh_last_error"}, raw
    biwi = open(os.path.join(csrc, "dh_biwi.cpp")).read()
    for name, body in re.findall(r'extern "C" int (dh_\w+)\([^)]*\) \{(.*?)\n\}', biwi, flags=re.S):
        assert "dh_guard_(" in body, name
        guarded.add(name)
    declared = set(declared_functions()) - {"dh_version", "dh_last_error"}
    assert declared == guarded, (sorted(declared - guarded), sorted(guarded - declared))
