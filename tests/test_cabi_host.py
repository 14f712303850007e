"""The C ABI from a host that is neither Python nor torch: tests/cabi_host.cpp (hipMalloc'd buffers, raw pointers, POD
descriptors, its own stream) drives a whole sigma call through liblanpaint_hip.so and compares with the scalar C
restatement of the reference linked into that test binary.  CPU suite: the program compiles against include/lanpaint_hip.h
and links against the library (every symbol it uses resolves).  GPU suite: it runs, VE and flow, several step counts."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _build(tmp_path):
    from lanpaint_amd import _cabi
    if not os.path.exists(_cabi.LIB_PATH):
        pytest.skip("liblanpaint_hip.so is not built")
    if not (shutil.which("g++") and shutil.which("gcc") and os.path.exists(os.path.join(ROCM, "include", "hip", "hip_runtime_api.h"))):
        pytest.skip("g++ / ROCm headers not available")
    obj = tmp_path / "langevin_oracle.o"
    subprocess.run(["gcc", "-c", "-O2", "-std=c11", "-ffp-contract=off", os.path.join(ROOT, "oracle", "langevin_oracle.c"),
                    "-o", str(obj)], check=True)
    exe = tmp_path / "cabi_host"
    lib_dir = os.path.dirname(_cabi.LIB_PATH)
    subprocess.run(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROCM, "include"), os.path.join(ROOT, "tests", "cabi_host.cpp"), str(obj),
                    "-L", lib_dir, "-llanpaint_hip", "-L", os.path.join(ROCM, "lib"), "-lamdhip64",
                    f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{os.path.join(ROCM, 'lib')}", "-o", str(exe)], check=True)
    return exe


def test_c_host_program_compiles_and_links_against_the_library(tmp_path):
    exe = _build(tmp_path)
    assert os.path.getsize(exe) > 0
    undefined = subprocess.run(["nm", "-D", "--undefined-only", str(exe)], capture_output=True, text=True, check=True).stdout
    used = {ln.split()[-1] for ln in undefined.splitlines() if " lp_" in ln}
    assert {"lp_abi_version", "lp_coeffs", "lp_step", "lp_finalize", "lp_strerror"} <= used


@pytest.mark.gpu
@pytest.mark.parametrize("flow,n_steps", [(0, 4), (1, 3), (0, 1), (1, 6)])
def test_c_host_program_matches_the_c_oracle(tmp_path, flow, n_steps):
    exe = _build(tmp_path)
    r = subprocess.run([str(exe), str(flow), str(n_steps)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "max|out - oracle|" in r.stdout
