"""Sanitizers run on the CPU build only: the library's host side under AddressSanitizer, in this container. This file is listed in
.gpurunignore: GPU boxes of this pool refuse sanitizer builds (GPU ASan needs xnack+ code objects, which are not available there),
and nothing in it needs a GPU."""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def test_host_entry_points_under_address_sanitizer(tmp_path):
    """Sanitizers run on the CPU build only (GPU ASan needs xnack+, which this pool does not offer): the library's host side compiled
    with -fsanitize=address (the flag is ignored for the gfx950 code object) and a driver that sweeps every host-only entry point."""
    import subprocess

    hipcc = "/opt/rocm/bin/hipcc"
    lib = tmp_path / "libtfft_asan.so"
    r = subprocess.run([hipcc, "-O1", "-g", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-slp-vectorize",
                        "-fsanitize=address", "-fno-omit-frame-pointer", "-o", str(lib),
                        os.path.join(ROOT, "tensor-fft_amd", "csrc", "tfft.hip"), "-ldl"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("this hipcc cannot build the host side with -fsanitize=address: " + r.stderr[-300:])
    exe = tmp_path / "asan_host"
    subprocess.check_call([hipcc, "-O1", "-g", "-std=c++17", "-fsanitize=address", "-fno-omit-frame-pointer", "-I",
                           os.path.join(ROOT, "include"), "-o", str(exe), os.path.join(ROOT, "tests", "cxx", "asan_host.cpp"),
                           "-L", str(tmp_path), "-ltfft_asan", "-Wl,-rpath," + str(tmp_path)], stderr=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr[-2000:]
    assert "AddressSanitizer" not in r.stderr
