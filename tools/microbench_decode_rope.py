"""Decode layer with the rotary embedding inside the attention launch (pe_decode_step_rope)
against the separate launches (rotary embedding of the step's rows, then pe_decode_step), per
layer, graph-replayed over several distinct caches so that nothing stays in the caches.

  python tools/microbench_decode_rope.py [B,Hq,Hkv,D,kv ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vllm_paged_compression_amd.ops as ops  # noqa: E402

SHAPES = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:] if "," in a] or [
    (64, 32, 8, 128, 776), (64, 32, 8, 128, 4104), (1, 32, 8, 128, 776), (8, 32, 8, 128, 776),
    (64, 8, 1, 128, 3080), (512, 4, 1, 128, 776)]


def time_graph(run, layers, iters=20):
    for i in range(layers):
        run(i)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        for i in range(layers):
            run(i)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for i in range(layers):
                run(i)
    graph.replay()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        graph.replay()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) * 1e3 / (iters * layers)


def main():
    dev = torch.device("cuda:0")
    bs, layers, max_pos = 16, 8, 8192
    print("B Hq Hkv D kv | separate (rope + decode_step) | fused, positions | fused, per-step table | decode_step alone  (us per layer)")
    for B, Hq, Hkv, D, kv in SHAPES:
        nblk = (kv + bs - 1) // bs
        nb = B * nblk + 8
        caches = [(torch.empty(nb, Hkv, D // 8, bs, 8, dtype=torch.bfloat16, device=dev).uniform_(-0.09, 0.09),
                   torch.empty(nb, Hkv, D, bs, dtype=torch.bfloat16, device=dev).uniform_(-0.09, 0.09))
                  for _ in range(layers)]
        qkv = [torch.empty(B, (Hq + 2 * Hkv) * D, dtype=torch.bfloat16, device=dev).uniform_(-0.09, 0.09)
               for _ in range(layers)]
        cs = torch.empty(max_pos, D, dtype=torch.bfloat16, device=dev).uniform_(-1, 1)
        pos = torch.randint(0, max_pos, (B,), device=dev, dtype=torch.int64)
        bt = torch.randperm(nb - 8, device=dev)[:B * nblk].to(torch.int32).reshape(B, nblk)
        lens = torch.full((B,), kv, dtype=torch.int32, device=dev)
        slots = (bt[:, (kv - 1) // bs].to(torch.int64) * bs + (kv - 1) % bs).contiguous()
        out = torch.empty(B, Hq, D, dtype=torch.bfloat16, device=dev)

        def views(i):
            q, k, v = qkv[i % layers].split([Hq * D, Hkv * D, Hkv * D], dim=-1)
            return q.view(B, Hq, D), k.view(B, Hkv, D), v.view(B, Hkv, D)

        def separate(i):
            q, k, v = views(i)
            kc, vc = caches[i % layers]
            ops.rotary_embedding(pos, q, k, D, cs, True)
            ops.decode_step(out, q, k, v, kc, vc, slots, Hkv, D ** -0.5, bt, lens, bs, kv)

        def fused(i):
            q, k, v = views(i)
            kc, vc = caches[i % layers]
            ops.decode_step_rope(out, pos, q, k, v, kc, vc, slots, Hkv, D ** -0.5, bt, lens, bs, kv, cs, True)

        step_cs = cs.index_select(0, pos)

        def fused_step_table(i):
            q, k, v = views(i)
            kc, vc = caches[i % layers]
            ops.decode_step_rope(out, None, q, k, v, kc, vc, slots, Hkv, D ** -0.5, bt, lens, bs, kv, step_cs, True)

        def plain(i):
            q, k, v = views(i)
            kc, vc = caches[i % layers]
            ops.decode_step(out, q, k, v, kc, vc, slots, Hkv, D ** -0.5, bt, lens, bs, kv)

        t_sep, t_fused, t_tab, t_plain = (time_graph(f, layers) for f in (separate, fused, fused_step_table, plain))
        print(f"{B} {Hq} {Hkv} {D} {kv} | {t_sep:8.2f} | {t_fused:8.2f} | {t_tab:8.2f} | {t_plain:8.2f}", flush=True)


if __name__ == "__main__":
    main()
