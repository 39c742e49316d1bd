"""Builds libpaged_evict.so (all HIP kernels + the C ABI) for gfx950, in-tree.

    python -m vllm_paged_compression_amd.build        (or)   python vllm-paged-compression_amd/build.py

hipcc cross-compiles without a GPU.  Objects go to build/ (git-ignored); the shared
library lands next to this file so that it travels with the source tree.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpaged_evict.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "paged_evict.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, obj):
    subprocess.check_call(["hipcc", *FLAGS, "-c", src, "-o", obj])
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hm = _headers_mtime()
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hm):
            jobs.append((src, obj))
    if jobs:
        if verbose:
            print("hipcc:", ", ".join(os.path.basename(s) for s, _ in jobs), flush=True)
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(lambda a: _compile(*a), jobs))
    if jobs or not os.path.exists(LIB):
        subprocess.check_call(["hipcc", "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB, *objs])
        if verbose:
            print("linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
