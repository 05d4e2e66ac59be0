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
SOURCES = ["ehx_api.cpp", "ehx_space.cpp", "ehx_flat.cpp", "ehx_graph.cpp", "ehx_shards.cpp", "ehx_write.cpp", "ehx_search.cpp",
           "k_flat.hip", "k_flat8.hip", "k_flat16.hip", "k_flati8.hip", "k_select.hip", "k_misc.hip",
           "k_graph.hip", "k_graphw.hip", "k_insert.hip"]
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
    defs = os.environ.get("EHX_DEFS", "").split()
    obj_dir = os.path.join(LIB_DIR, "obj%s" % os.environ.get("EHX_LIB_SUFFIX", ""))
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [
        os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    newest_header = max(os.path.getmtime(h) for h in headers)
    stamp = os.path.join(obj_dir, "defs.txt")  # objects built with other -D flags are not reused
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(defs):
        force = True
    compile_flags = [f for f in FLAGS if f != "-shared"]

    def compile_one(src):
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header):
            return obj
        cmd = [HIPCC] + compile_flags + defs + ["-x", "hip", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        return obj
    # one hipcc per translation unit, side by side (the kernels are independent files; -fno-gpu-rdc)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, srcs))
    with open(stamp, "w") as f:
        f.write(" ".join(defs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
