"""Diagnostic build of the decode kernel with per-workgroup time stamps (100 MHz constant clock):
kernel entry, first tile landed, main loop done, kernel exit.  Writes tools/_exp/libpe_stamps.so,
which also exports pe_debug_read_stamps(dst, n)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vllm-paged-compression_amd")
SRC = os.path.join(PKG, "csrc", "paged_decode.hip")
OUT = os.path.join(ROOT, "tools", "_exp")


def rep(text, old, new):
    assert old in text, old
    return text.replace(old, new, 1)


def main():
    os.makedirs(OUT, exist_ok=True)
    t = open(SRC).read()
    t = rep(t, "namespace pe {\n", "namespace pe {\n__device__ unsigned long long g_stamps[16384 * 4];\n")
    t = rep(t, "  const int split = blockIdx.x;\n",
            "  const unsigned long long st0 = __builtin_amdgcn_s_memrealtime();\n"
            "  unsigned long long st1 = st0, st2 = st0;\n  const int split = blockIdx.x;\n")
    t = rep(t, "      issue(t0, p);\n      while (true) {\n",
            "      issue(t0, p);\n      asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n"
            "      st1 = __builtin_amdgcn_s_memrealtime();\n      while (true) {\n")
    t = rep(t, "  // ---- the step's new token:", "  st2 = __builtin_amdgcn_s_memrealtime();\n  // ---- the step's new token:")
    # end of kernel: after the final output loop
    t = rep(t, "      w[2 + d] = o;\n    }\n  }\n}\n",
            "      w[2 + d] = o;\n    }\n  }\n"
            "  if (threadIdx.x == 0) {\n"
            "    const int wg = ((int)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;\n"
            "    if (wg < 16384) {\n"
            "      g_stamps[wg * 4 + 0] = st0; g_stamps[wg * 4 + 1] = st1; g_stamps[wg * 4 + 2] = st2;\n"
            "      g_stamps[wg * 4 + 3] = __builtin_amdgcn_s_memrealtime();\n    }\n  }\n}\n")
    t += ("\nextern \"C\" int pe_debug_read_stamps(unsigned long long* dst, int n) {\n"
          "  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pe::g_stamps), (size_t)n * 8);\n}\n")
    src = os.path.join(PKG, "csrc", "_exp_stamps.hip")
    with open(src, "w") as f:
        f.write(t)
    try:
        obj = os.path.join(OUT, "decode_stamps.o")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                               "-I" + os.path.join(ROOT, "include"), "-c", src, "-o", obj])
        objs = [os.path.join(PKG, "build", f) for f in os.listdir(os.path.join(PKG, "build"))
                if f.endswith(".o") and f != "paged_decode.o"]
        lib = os.path.join(OUT, "libpe_stamps.so")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj, *objs])
        os.remove(obj)
        print("built", lib)
    finally:
        os.remove(src)


if __name__ == "__main__":
    main()
