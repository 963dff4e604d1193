"""CPU-side sanitizer run of the library's host code (GPU sanitizers are not available on the pool):
romtime_amd/csrc/host_dense.cpp - the Jacobi eigensolver behind rt_host_jacobi_eigh, the k x k generalised eigenproblem of
the Rayleigh-Ritz step and the truncation rule of rt_pod_orth - is compiled with g++ -fsanitize=address,undefined together
with tests/host/host_dense_check.cpp and executed."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_dense_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    exe = str(tmp_path / "host_dense_check")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           os.path.join(REPO, "tests", "host", "host_dense_check.cpp"), os.path.join(REPO, "romtime_amd", "csrc", "host_dense.cpp"),
           "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0, run.stdout + run.stderr
    assert "host_dense_check ok" in run.stdout
