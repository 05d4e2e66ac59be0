"""Builds the engine's shared library for gfx950 with hipcc (in-tree, so the .so travels to the
GPU box with the repo snapshot).  Used by __graft_entry__.build() and by developers."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libehx%s.so" % os.environ.get("EHX_LIB_SUFFIX", ""))  # suffix: ablation builds
SOURCES = ["ehx_api.cpp", "k_flat.hip", "k_flat8.hip", "k_flat16.hip", "k_flati8.hip", "k_select.hip", "k_misc.hip",
           "k_graph.hip", "k_insert.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
         # host side only: F16C/AVX2 for the fp32 -> binary16 conversion of rows written to fp16 spaces (every
         # x86 host of an MI355X has them; without F16C the conversion is a software routine, ~10x slower)
         "-Xarch_host", "-mf16c", "-Xarch_host", "-mavx2"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIB_DIR, exist_ok=True)
    if not force and not _stale():
        return LIB
    srcs = [os.path.join(CSRC, f) for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    cmd = [HIPCC] + FLAGS + os.environ.get("EHX_DEFS", "").split() + ["-x", "hip"] + srcs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
