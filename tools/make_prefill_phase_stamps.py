"""Diagnostic build of the prefill kernel that accumulates, per wave, the shader-clock cycles
spent in the phases of the tile loop (QK, softmax, P.V, wait for the staged tile, barrier) and
exports pe_debug_read_pf_phases(dst, n).  Writes tools/_exp/libpe_pfphase.so."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vllm-paged-compression_amd")
SRC = os.path.join(PKG, "csrc", "prefill.hip")
OUT = os.path.join(ROOT, "tools", "_exp")


def rep(text, old, new, count=1):
    assert old in text, old
    return text.replace(old, new, count)


def main():
    os.makedirs(OUT, exist_ok=True)
    t = open(SRC).read()
    t = rep(t, "namespace pe {\n", "namespace pe {\n__device__ unsigned long long g_pf_phase[8192 * 4 * 8];\n"
            "#define PF_NOW() ([] { __builtin_amdgcn_sched_barrier(0); unsigned long long x = __builtin_amdgcn_s_memtime();"
            " __builtin_amdgcn_sched_barrier(0); return x; }())\n")
    t = rep(t, "  auto compute = [&](int tile, int buf, auto masked_tag) {\n",
            "  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};\n"
            "  auto compute = [&](int tile, int buf, auto masked_tag) {\n"
            "    const unsigned long long c0 = PF_NOW();\n")
    t = rep(t, "    // ---- mask, online softmax (the scale rides", "    const unsigned long long c1 = PF_NOW();\n"
            "    // ---- mask, online softmax (the scale rides")
    t = rep(t, "    // ---- O^T += V^T.P^T ----", "    const unsigned long long c2 = PF_NOW();\n    // ---- O^T += V^T.P^T ----")
    t = rep(t, "        for (int n = 0; n < NR; ++n) acc[dt][n] = TR::mfma16(vf, pf[n][k2], acc[dt][n]);\n      }\n    }\n  };\n",
            "        for (int n = 0; n < NR; ++n) acc[dt][n] = TR::mfma16(vf, pf[n][k2], acc[dt][n]);\n      }\n    }\n"
            "    const unsigned long long c3 = PF_NOW();\n"
            "    ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += c3 - c2; ph[6] += 1;\n  };\n")
    t = rep(t, "      if (tile + 1 < n_tiles) stage(tile + 1, buf ^ 1);\n",
            "      const unsigned long long l0 = PF_NOW();\n"
            "      if (tile + 1 < n_tiles) stage(tile + 1, buf ^ 1);\n"
            "      const unsigned long long l1 = PF_NOW();\n")
    t = rep(t, "      stage_wait();\n      __syncthreads();\n    }\n  };\n",
            "      const unsigned long long l2 = PF_NOW();\n      stage_wait();\n"
            "      const unsigned long long l3 = PF_NOW();\n      __syncthreads();\n"
            "      const unsigned long long l4 = PF_NOW();\n"
            "      ph[3] += l1 - l0; ph[4] += l3 - l2; ph[5] += l4 - l3; ph[7] += l4 - l0;\n    }\n  };\n")
    t = rep(t, "          u32x4{lo[0], lo[1], hi[0], hi[1]};\n    }\n  }\n}\n",
            "          u32x4{lo[0], lo[1], hi[0], hi[1]};\n    }\n  }\n"
            "  if (lane == 0) {\n"
            "    const int wg = ((int)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;\n"
            "    if (wg < 8192)\n"
            "      for (int i = 0; i < 8; ++i) g_pf_phase[(wg * 4 + wave) * 8 + i] = ph[i];\n  }\n}\n")
    t += ("\nextern \"C\" int pe_debug_read_pf_phases(unsigned long long* dst, int n) {\n"
          "  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pe::g_pf_phase), (size_t)n * 8);\n}\n")
    src = os.path.join(OUT, "pfphase.hip")
    with open(src, "w") as f:
        f.write(t)
    obj = os.path.join(OUT, "pfphase.o")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"),
                           "-c", src, "-o", obj])
    objs = [os.path.join(PKG, "build", f) for f in os.listdir(os.path.join(PKG, "build"))
            if f.endswith(".o") and f != "prefill.o"]
    lib = os.path.join(OUT, "libpe_pfphase.so")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj, *objs])
    os.remove(obj)
    os.remove(src)
    print("built", lib)


if __name__ == "__main__":
    main()
