"""The C++ host-side mirror of the reference interface (knoxdb_amd/csrc/host/knox_host.hpp):
builds against the C ABI on CPU; the GPU run executes tests/cpp/test_host.cpp, which reads like
the reference's match_test.go / bitset_test.go / PhysicalFilter usage."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_host")
GO_EXE = os.path.join(ROOT, "tests", "cpp", "test_go_entries")


def _build():
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", os.path.join(ROOT, "tests", "cpp", "test_host.cpp"),
           "-I" + os.path.join(ROOT, "knoxdb_amd", "csrc", "host"), "-L" + os.path.join(ROOT, "knoxdb_amd"), "-lknoxgpu",
           "-Wl,-rpath," + os.path.join(ROOT, "knoxdb_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", EXE]
    subprocess.check_call(cmd)


def test_host_mirror_compiles_against_the_abi():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_host_mirror_runs_on_gpu():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host mirror ok" in out.stdout


def _build_go_entries():
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", os.path.join(ROOT, "tests", "cpp", "test_go_entries.cpp"),
           "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "knoxdb_amd"), "-lknoxgpu",
           "-Wl,-rpath," + os.path.join(ROOT, "knoxdb_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", GO_EXE]
    subprocess.check_call(cmd)


def test_go_entry_mirror_compiles_against_the_header():
    """tests/cpp/test_go_entries.cpp makes the C calls of integration/go/knoxgpu/knoxgpu.go, wrapper by wrapper"""
    _build_go_entries()
    assert os.path.exists(GO_EXE)


@pytest.mark.gpu
def test_go_entry_mirror_runs_on_gpu():
    _build_go_entries()
    out = subprocess.run([GO_EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "go entries ok" in out.stdout
