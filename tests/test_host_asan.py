"""CPU: the host-side sanitizer run (SURVEY.md 5 "race detection / sanitizers").  `make host-asan` builds
the HOST pass of every csrc/*.hip under AddressSanitizer + UBSan against tools/hipstub (the HIP runtime on host memory,
kernel launches are no-ops); tools/host_asan_driver.py then runs checkpoint folding / packing for all model kinds, weight
regions, the micro-batch planner over BASELINE configs[4]'s lengths, chunk planning, f0 files, error paths and the FLAC codec
on that library.  Any report aborts the run.  CPU container only: never on the GPU box (the gpu marker keeps it off there)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_asan_and_ubsan():
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not (os.path.exists(clang) and shutil.which("hipcc") and shutil.which("make")):
        pytest.skip("no ROCm clang / hipcc / make here")
    rt = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(rt):
        pytest.skip("no shared ASan runtime")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "host_asan.sh")], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    assert r.returncode == 0 and "HOST_ASAN_OK" in r.stdout, tail
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, tail
