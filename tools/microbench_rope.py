"""Times the fused rotary-embedding + KV-write kernel against its two-launch form (rotation only,
then pe_reshape_and_cache) and reports algorithmic HBM bandwidth."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vllm_paged_compression_amd.ops as ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=8192)
    ap.add_argument("--hq", type=int, default=32)
    ap.add_argument("--hkv", type=int, default=8)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    T, Hq, Hkv, D, bs = a.tokens, a.hq, a.hkv, a.d, 16
    nb = (T + bs - 1) // bs + 4
    n_buf = 8   # rotate over several buffers so that nothing stays in the caches
    qkvs = [torch.empty(T, (Hq + 2 * Hkv) * D, dtype=torch.bfloat16, device=dev).uniform_(-1, 1) for _ in range(n_buf)]
    caches = [(torch.zeros(nb, Hkv, D // 8, bs, 8, dtype=torch.bfloat16, device=dev),
               torch.zeros(nb, Hkv, D, bs, dtype=torch.bfloat16, device=dev)) for _ in range(n_buf)]
    cs = torch.empty(T + 16, D, dtype=torch.bfloat16, device=dev).uniform_(-1, 1)
    pos = torch.arange(T, device=dev, dtype=torch.int64)
    slots = torch.arange(T, device=dev, dtype=torch.int64)

    def fused(i):
        q, k, v = qkvs[i % n_buf].split([Hq * D, Hkv * D, Hkv * D], dim=-1)
        kc, vc = caches[i % n_buf]
        ops.rotary_embedding_and_cache(pos, q, k, v, kc, vc, slots, D, cs, True)

    def split(i):
        q, k, v = qkvs[i % n_buf].split([Hq * D, Hkv * D, Hkv * D], dim=-1)
        kc, vc = caches[i % n_buf]
        ops.rotary_embedding(pos, q, k, D, cs, True)
        ops.reshape_and_cache(k.view(T, Hkv, D), v.view(T, Hkv, D), kc, vc, slots)

    nbytes = T * D * 2 * (2 * Hq + 2 * Hkv + Hkv + 2 * Hkv)   # Q r+w, K r+w, K cache w, V r + cache w
    for name, fn in (("fused", fused), ("rotation + KV write", split)):
        for i in range(n_buf):
            fn(i)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for i in range(a.iters):
            fn(i)
        en.record()
        torch.cuda.synchronize()
        us = st.elapsed_time(en) * 1e3 / a.iters
        print(f"{name:22s} T={T}: {us:8.2f} us  {nbytes / us / 1e6:.3f} TB/s algorithmic (fused form, "
              f"{nbytes / 1e6:.1f} MB)", flush=True)


if __name__ == "__main__":
    main()
