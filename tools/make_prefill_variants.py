"""Builds experimental variants of the prefill kernel (wrong results, timing only) into tools/_exp/.

Each variant removes one ingredient of the main loop so that its cost can be read off the
difference in kernel time:  noload (no global->LDS staging after tile 0), nobar (noload + no
barrier), nosoftmax (no exp2), novalu (no max / exp2 / rescale), nopv (no P.V MFMAs).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vllm-paged-compression_amd")
SRC = os.path.join(PKG, "csrc", "prefill.hip")
OUT = os.path.join(ROOT, "tools", "_exp")


def variant(name, text):
    if name in ("noload", "nobar"):
        text = text.replace("if (tile + 1 < n_tiles) stage(tile + 1, buf ^ 1);", "")
        text = text.replace("const int buf = tile & 1;", "const int buf = 0;")
    if name == "nobar":
        text = text.replace("      stage_wait();\n      __syncthreads();\n    }\n  };", "    }\n  };")
    if name == "nosoftmax":
        text = text.replace(
            "const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[sub][n][r], scale_log2e, -m_new));",
            "const float p = s[sub][n][r];")
    if name == "novalu":
        # no max, no exp, no rescale: P is the raw S (wrong numbers, same MFMA / LDS work)
        text = text.replace("for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[sub][n][r]);", "for (int r = 0; r < 1; ++r) mx = fmaxf(mx, s[sub][n][r]);")
        text = text.replace(
            "const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[sub][n][r], scale_log2e, -m_new));",
            "const float p = s[sub][n][r];")
        text = text.replace("if (__builtin_amdgcn_ballot_w64(m_new > m_run[n]) != 0) {", "if (m_new > 1e30f) {")
    if name in ("nopv", "nopv_novalu", "qkonly"):
        text = text.replace("for (int n = 0; n < NR; ++n) acc[dt][n] = TR::mfma16(vf, pf[n][k2], acc[dt][n]);",
                            "for (int n = 0; n < NR; ++n) acc[dt][n][0] += (float)vf[n] + (float)pf[n][k2][dt & 7];")
    if name in ("nopv_novalu", "qkonly"):
        text = variant("novalu", text)
    if name == "qkonly":     # + no staging, no barrier: QK MFMAs, K and V reads only
        text = variant("nobar", text)
    if name == "qk_nov":     # QK MFMAs and K reads only: no V reads, no softmax, no P.V, no staging, no barrier
        text = variant("qkonly", text)
        a = text.index("    const int tq = c >> 2, tp = c & 3;")
        b = text.index("  };\n\n  // ---- main loop")
        text = text[:a] + ("#pragma unroll\n    for (int sub = 0; sub < 4; ++sub)\n#pragma unroll\n"
                           "      for (int n = 0; n < NR; ++n) acc[sub][n] += s[sub][n];\n"
                           "    (void)pf;\n") + text[b:]
    if name == "qk_noread":  # qk_nov without the K reads: MFMAs on register operands only
        text = variant("qk_nov", text)
        text = text.replace("""        dst[ks] = __builtin_bit_cast(
            vec8, *reinterpret_cast<const u32x4*>(kt + swz_off<D>(sub * 16 + c, 4 * ks + g)));""",
                            "        dst[ks] = qf[0][(ks + sub) % KS];")
    if name == "qk_noread32":  # the same flops as qk_noread issued as 16 v_mfma_f32_32x32x16
        text = variant("qk_nov", text)
        a = text.index("    f32x4 s[4][NR];")
        b = text.index("    // ---- mask, online softmax")
        text = text[:a] + """    f32x4 s[4][NR];
    f32x16 z[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) z[j][i] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        z[j] = TR::mfma32(qf[j % NR][i % KS], qf[(j + 1) % NR][(i + 1) % KS], z[j]);
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
      for (int n = 0; n < NR; ++n)
        s[sub][n] = f32x4{z[n & 1][4 * sub], z[n & 1][4 * sub + 1], z[n & 1][4 * sub + 2], z[n & 1][4 * sub + 3]};
    (void)kt;
""" + text[b:]
    if name == "skeleton":   # qk_noread without the MFMAs: what the tile loop itself costs
        text = variant("qk_noread", text)
        text = text.replace("s[sub][n] = TR::mfma16(kf[sub & 1][ks], qf[n][ks], s[sub][n]);",
                            "s[sub][n][ks & 3] += (float)kf[sub & 1][ks][n];")
    if name == "v_noqk":     # V reads + P.V MFMAs only (P = ones), no K reads, no QK, no softmax, no staging, no barrier
        text = variant("nobar", text)
        a = text.index("    f32x4 s[4][NR];")
        b = text.index("    // ---- O^T += V^T.P^T")
        text = text[:a] + ("    vec8 pf[NR][2];\n#pragma unroll\n    for (int n = 0; n < NR; ++n) { pf[n][0] = ones; pf[n][1] = ones; }\n"
                           "    (void)n0; (void)kt;\n") + text[b:]
    return text


def main():
    os.makedirs(OUT, exist_ok=True)
    base = open(SRC).read()
    objs = [os.path.join(PKG, "build", f) for f in os.listdir(os.path.join(PKG, "build"))
            if f.endswith(".o") and f != "prefill.o"]
    for name in sys.argv[1:] or ["noload", "nobar", "nosoftmax", "novalu", "nopv"]:
        text = variant(name, base)
        assert text != base, name
        src = os.path.join(PKG, "csrc", f"_exp_{name}.hip")
        with open(src, "w") as f:
            f.write(text)
        try:
            lib = os.path.join(OUT, f"libpe_{name}.so")
            obj = os.path.join(OUT, f"prefill_{name}.o")
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                                   "-I" + os.path.join(ROOT, "include"), "-c", src, "-o", obj])
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj, *objs])
            os.remove(obj)
            print("built", lib)
        finally:
            os.remove(src)


if __name__ == "__main__":
    main()
