"""The host-only logic of the runtime (depthhead_amd/csrc/dh_host.cpp, dh_biwi.cpp: forest validation, knob parsing, tile
selection, upload chunk plans, run-length payload validation / packing, the exception guard, dh_parallel_for_) under the CPU
sanitizers: tests/host/host_check.cpp built with g++ -fsanitize=address,undefined and once more with -fsanitize=thread.
(GPU AddressSanitizer is not available on the pool; sanitizers run on the CPU build only.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "depthhead_amd", "csrc")
SOURCES = [os.path.join(ROOT, "tests", "host", "host_check.cpp"), os.path.join(CSRC, "dh_host.cpp"), os.path.join(CSRC, "dh_biwi.cpp")]


def _build_and_run(tmp_path, sanitize, env_extra):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "host_check")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off", f"-fsanitize={sanitize}",
           "-fno-sanitize-recover=undefined", "-pthread", *SOURCES, "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0 and ("cannot find -l" in res.stderr or "unrecognized" in res.stderr):
        pytest.skip(f"sanitizer runtime for {sanitize} not installed: {res.stderr[-200:]}")
    assert res.returncode == 0, res.stderr[-3000:]
    env = dict(os.environ, **env_extra)
    for k in [k for k in env if k.startswith("DH_")]:
        env.pop(k)
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0 and "host_check ok" in run.stdout, (run.stdout[-2000:], run.stderr[-6000:])


def test_host_logic_under_asan_ubsan(tmp_path):
    _build_and_run(tmp_path, "address,undefined", {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})


def test_host_logic_under_tsan(tmp_path):
    _build_and_run(tmp_path, "thread", {"HOST_CHECK_LIGHT": "1", "TSAN_OPTIONS": "halt_on_error=1"})


def test_host_translation_units_have_no_hip_in_them():
    """dh_host.cpp / dh_host.h / dh_biwi.cpp stay buildable by a plain C++ compiler: no HIP header, type or call."""
    for fn in ("dh_host.cpp", "dh_host.h", "dh_biwi.cpp"):
        txt = open(os.path.join(CSRC, fn)).read()
        for word in ("hip/hip_runtime", "hipMalloc", "hipStream", "hipError_t", "__global__", "__device__", "dh_internal.h"):
            assert word not in txt, (fn, word)
