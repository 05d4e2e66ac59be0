"""Builds integration/cpp/index_test.cc (the reference's index_test.cc:17-60 against the drop-in
ANNIndex of integration/cpp/ann_index.h) with g++ and runs it on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "integration", "cpp", "index_test.bin")


def build():
    lib_dir = os.path.join(ROOT, "embeddinghub_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "integration", "cpp", "index_test.cc"), "-o", BIN,
                           "-L", lib_dir, "-lehx", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    return BIN


def test_dropin_header_compiles_and_links():
    import __graft_entry__ as graft
    graft.build()
    assert os.path.exists(build())


@pytest.mark.gpu
def test_reference_index_tests_pass_on_the_engine():
    if not os.path.exists(BIN):
        build()
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "4 passed" in out.stdout
