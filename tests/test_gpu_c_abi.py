"""The C ABI from plain C: compile tests/c_abi/smoke.c with gcc against include/wgsassign_hip.h and
libwgsassign_hip.so and run it."""
import os
import subprocess

import pytest

from conftest import ROOT


def _compile(tmp_path):
    exe = str(tmp_path / "c_smoke")
    lib_dir = os.path.join(ROOT, "wgsassign_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_abi", "smoke.c"), "-o", exe, "-L", lib_dir, "-lwgsassign_hip", "-lm",
                    "-Wl,-rpath," + lib_dir], check=True, capture_output=True, text=True)
    return exe


def test_header_compiles_as_c99(tmp_path):
    """CPU: the public header is valid C99 and the library links from C."""
    from wgsassign_amd import build
    build.build()
    assert os.path.exists(_compile(tmp_path))


@pytest.mark.gpu
def test_c_caller_runs(tmp_path):
    r = subprocess.run([_compile(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "C ABI smoke OK" in r.stdout
