"""The C++ host class (cmc_fluid_solver_amd/host/AdiSolver3D_hip.h, the reference-shaped Solver3D
interface over the C ABI): compiles everywhere, runs against the oracle on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "host_driver_test")


def _build():
    cmd = ["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "host_driver_test.cpp"), "-o", EXE,
           "-L" + os.path.join(ROOT, "cmc_fluid_solver_amd"), "-lfs3d_hip", "-L" + os.path.join(ROOT, "oracle"), "-loracle",
           "-Wl,-rpath," + os.path.join(ROOT, "cmc_fluid_solver_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-fopenmp"]
    subprocess.check_call(cmd)


def test_host_class_compiles_and_links(built):
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_host_class_matches_oracle(built):
    _build()
    # the C++ test compares with the oracle bit for bit: contexts start on the bit-exact kernels (tolerance tests of the
    # partition kernels: tests/test_gpu_part.py)
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300, env=dict(os.environ, FS3D_DEFAULT_KERNEL="4"))
    assert "HOST_CPP_OK" in out.stdout, out.stdout + out.stderr
