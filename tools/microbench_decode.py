"""Times the decode-attention kernel alone (HIP events on the launch stream)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vllm_paged_compression_amd.ops as ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--kv", type=int, default=776)
    ap.add_argument("--hq", type=int, default=32)
    ap.add_argument("--hkv", type=int, default=8)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--fused", action="store_true", help="time pe_decode_step (KV write fused)")
    ap.add_argument("--sequential", action="store_true", help="block tables in address order")
    ap.add_argument("--lib", default=None, help="alternative build of the library (A/B timing)")
    ap.add_argument("--layers", type=int, default=32, help="distinct caches cycled through (defeats L2/MALL reuse)")
    a = ap.parse_args()
    if a.lib:
        import vllm_paged_compression_amd._lib as _lib
        _lib.LIB_PATH = os.path.abspath(a.lib)
    dev = torch.device("cuda:0")
    bs = 16
    nblk_seq = (a.kv + bs - 1) // bs
    nb = a.batch * nblk_seq + 8
    torch.manual_seed(0)
    caches = []
    for _ in range(a.layers):
        kc = torch.empty(nb, a.hkv, a.d // 8, bs, 8, dtype=torch.bfloat16, device=dev).uniform_(-0.09, 0.09)
        vc = torch.empty(nb, a.hkv, a.d, bs, dtype=torch.bfloat16, device=dev).uniform_(-0.09, 0.09)
        caches.append((kc, vc))
    q = torch.empty(a.batch, a.hq, a.d, dtype=torch.bfloat16, device=dev).uniform_(-0.09, 0.09)
    bt = torch.randperm(nb - 8, device=dev)[:a.batch * nblk_seq].to(torch.int32).reshape(a.batch, nblk_seq)
    if a.sequential:
        bt = torch.arange(a.batch * nblk_seq, device=dev, dtype=torch.int32).reshape(a.batch, nblk_seq)
    lens = torch.full((a.batch,), a.kv, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    scale = a.d ** -0.5

    k_new = torch.empty(a.batch, a.hkv, a.d, dtype=torch.bfloat16, device=dev).uniform_(-0.09, 0.09)
    v_new = torch.empty(a.batch, a.hkv, a.d, dtype=torch.bfloat16, device=dev).uniform_(-0.09, 0.09)
    slots = bt[:, (a.kv - 1) // bs].to(torch.int64) * bs + (a.kv - 1) % bs

    launcher = None
    if a.fused:
        kvs = [torch.stack([kc.view(nb, -1), vc.view(nb, -1)]) for kc, vc in caches]
        caches = None
        launcher = ops.DecodeLayerLauncher(q.view(a.batch, -1), k_new.view(a.batch, -1),
                                           v_new.view(a.batch, -1), kvs[0], a.hq, a.hkv, a.d, slots,
                                           None, bt, lens, a.kv, scale)
        assert launcher.fused
        q2, k2, v2, o2 = q.view(a.batch, -1), k_new.view(a.batch, -1), v_new.view(a.batch, -1), out.view(a.batch, -1)

    def run(i):
        if launcher is not None:
            launcher.run(q2, k2, v2, kvs[i % a.layers], o2)
            return
        kc, vc = caches[i % a.layers]
        if False:
            pass
        else:
            ops.paged_attention_v1(out, q, kc, vc, a.hkv, scale, bt, lens, bs, a.kv, None, "auto", 1.0, 1.0)

    for i in range(a.layers):
        run(i)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    n = a.iters * a.layers
    for i in range(n):
        run(i)
    en.record()
    torch.cuda.synchronize()
    us = st.elapsed_time(en) * 1e3 / n
    bytes_ = a.batch * (2 * a.kv * a.hkv * a.d * 2 + 2 * a.hq * a.d * 2 + nblk_seq * 4)
    print(f"decode B={a.batch} kv={a.kv} Hq={a.hq} Hkv={a.hkv} D={a.d}: {us:.2f} us/launch, "
          f"{bytes_ / us / 1e6:.3f} TB/s algorithmic ({bytes_/1e6:.1f} MB)", flush=True)


if __name__ == "__main__":
    main()
