"""the C ABI from C: tests/c_abi/driver.c is compiled with gcc against include/d3d_hip.h + the HIP runtime and run -- no
Python, no torch between the caller and libd3d_hip.so (the boundary a C++ host of the reference would use)"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_c_program_against_the_oracle(tmp_path):
    exe = str(tmp_path / "driver")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    subprocess.check_call(["gcc", "-std=c99", "-O1", os.path.join(ROOT, "tests", "c_abi", "driver.c"), "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(rocm, "include"), "-D__HIP_PLATFORM_AMD__", "-L" + os.path.join(ROOT, "d3d_amd"), "-ld3d_hip",
                           "-L" + os.path.join(ROOT, "oracle"), "-loracle", "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "d3d_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
                           "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all comparisons passed" in r.stdout and r.stdout.count("bit-exact") == 4
