"""The C ABI without Python or torch: a plain C host program linked against libpioneer_amd.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    """Plain gcc, C11: the header and the program are C, not C++/HIP."""
    exe = str(tmp_path / "abi_smoke")
    lib_dir = os.path.join(ROOT, "pioneer_amd", "csrc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
                    "-I" + os.path.join(rocm, "include"), "-I" + os.path.join(ROOT, "include"),
                    "-L" + lib_dir, "-lpioneer_amd", "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-lm",
                    "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", exe],
                   check=True, capture_output=True)
    return exe


def test_c_program_compiles_and_links(tmp_path, hip_lib):
    """(CPU) the header is valid C and every symbol the program uses resolves against the library."""
    exe = _build(tmp_path)
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_c_program_known_answers(tmp_path, hip_lib):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "abi_smoke ok" in out.stdout, out.stdout + out.stderr
