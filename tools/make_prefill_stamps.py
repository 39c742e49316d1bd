"""Diagnostic build of the prefill kernel with per-workgroup time stamps (100 MHz constant clock):
kernel entry, first tile landed, main loop done, exit, plus the CU the workgroup ran on.
Writes tools/_exp/libpe_pfstamps.so, which also exports pe_debug_read_pf_stamps(dst, n)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vllm-paged-compression_amd")
SRC = os.path.join(PKG, "csrc", "prefill.hip")
OUT = os.path.join(ROOT, "tools", "_exp")


def rep(text, old, new):
    assert old in text, old
    return text.replace(old, new, 1)


def main():
    os.makedirs(OUT, exist_ok=True)
    t = open(SRC).read()
    t = rep(t, "namespace pe {\n", "namespace pe {\n__device__ unsigned long long g_pf_stamps[8192 * 6];\n")
    t = rep(t, "  const int gqa = num_heads / num_kv_heads;\n  const int lid =",
            "  const unsigned long long st0 = __builtin_amdgcn_s_memrealtime();\n"
            "  unsigned long long st1 = st0, st2 = st0;\n"
            "  const int gqa = num_heads / num_kv_heads;\n  const int lid =")
    t = rep(t, "    stage_wait();\n    __syncthreads();\n    run_tiles(0, n_full, std::false_type{});\n"
               "    run_tiles(n_full, n_tiles, std::true_type{});\n",
            "    stage_wait();\n    __syncthreads();\n    st1 = __builtin_amdgcn_s_memrealtime();\n"
            "    run_tiles(0, n_full, std::false_type{});\n"
            "    run_tiles(n_full, n_tiles, std::true_type{});\n"
            "    st2 = __builtin_amdgcn_s_memrealtime();\n")
    t = rep(t, "          u32x4{lo[0], lo[1], hi[0], hi[1]};\n    }\n  }\n}\n",
            "          u32x4{lo[0], lo[1], hi[0], hi[1]};\n    }\n  }\n"
            "  if (threadIdx.x == 0) {\n"
            "    const int wg = ((int)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;\n"
            "    if (wg < 8192) {\n"
            "      unsigned long long* s = g_pf_stamps + wg * 6;\n"
            "      s[0] = st0; s[1] = st1; s[2] = st2; s[3] = __builtin_amdgcn_s_memrealtime();\n"
            "      s[4] = __builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 4)  /* HW_ID */\n"
            "             | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 11) | (0 << 6) | 4) << 32);  /* XCC_ID */\n"
            "      s[5] = n_tiles;\n    }\n  }\n}\n")
    t += ("\nextern \"C\" int pe_debug_read_pf_stamps(unsigned long long* dst, int n) {\n"
          "  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pe::g_pf_stamps), (size_t)n * 8);\n}\n")
    src = os.path.join(PKG, "csrc", "_exp_pfstamps.hip")
    with open(src, "w") as f:
        f.write(t)
    try:
        obj = os.path.join(OUT, "pfstamps.o")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                               "-I" + os.path.join(ROOT, "include"), "-c", src, "-o", obj])
        objs = [os.path.join(PKG, "build", f) for f in os.listdir(os.path.join(PKG, "build"))
                if f.endswith(".o") and f != "prefill.o"]
        lib = os.path.join(OUT, "libpe_pfstamps.so")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj, *objs])
        os.remove(obj)
        print("built", lib)
    finally:
        os.remove(src)


if __name__ == "__main__":
    main()
