"""Builds the library with other build-time tunables of the decode kernel into tools/_exp/.

usage: make_decode_variants.py name:-DPE_DEC_TILES=1,-DPE_DEC_OCC=4 ...
Each variant is timed against the default build on the same device with
tools/microbench_decode.py --lib tools/_exp/libpe_<name>.so
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vllm-paged-compression_amd")
SRC = os.path.join(PKG, "csrc", "paged_decode.hip")
OUT = os.path.join(ROOT, "tools", "_exp")


def main():
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(PKG, "build", f) for f in os.listdir(os.path.join(PKG, "build"))
            if f.endswith(".o") and f != "paged_decode.o"]
    for spec in sys.argv[1:]:
        name, _, flags = spec.partition(":")
        flags = [f for f in flags.split(",") if f]
        lib = os.path.join(OUT, f"libpe_{name}.so")
        obj = os.path.join(OUT, f"paged_decode_{name}.o")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                               "-I" + os.path.join(ROOT, "include"), *flags, "-c", SRC, "-o", obj,
                               "-Rpass-analysis=kernel-resource-usage"],
                              stderr=open(os.path.join(OUT, f"{name}.remarks"), "w"))
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj, *objs])
        os.remove(obj)
        rem = open(os.path.join(OUT, f"{name}.remarks")).read().split("Function Name")
        for r in rem:
            if "paged_decode_mfma_kernelIDF16bLi128" in r:
                keep = [l.split("remark:")[1].strip() for l in r.splitlines()
                        if any(k in l for k in (" VGPRs:", "Occupancy", "VGPRs Spill", "LDS Size"))]
                print(name, "|", "; ".join(keep))
        print("built", lib)


if __name__ == "__main__":
    main()
