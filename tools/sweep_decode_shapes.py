"""Sweep of the decode kernel's launch shape (waves per workgroup x KV splits) over batch shapes:
prints us per launch for every combination (the library re-reads PE_DECODE_WAVES / PE_DECODE_SPLITS
after ops.reload_env(), so one process covers the sweep).  '0/0' is the library's own choice."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vllm_paged_compression_amd.ops as ops  # noqa: E402

# (batch, Hq, Hkv, D, kv)
SHAPES = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:] if "," in a and "/" not in a] or [
    (1, 32, 8, 128, 776), (1, 12, 12, 64, 1074), (1, 32, 8, 128, 4104), (1, 32, 8, 128, 16400),
    (4, 32, 8, 128, 776), (8, 32, 8, 128, 776), (8, 32, 8, 128, 4104), (16, 32, 8, 128, 776),
    (64, 8, 1, 128, 3080), (64, 8, 1, 128, 776), (8, 8, 1, 128, 3080), (32, 32, 8, 128, 776),
    (64, 32, 8, 128, 776)]
COMBOS = [tuple(int(x) for x in a.split("/")) for a in sys.argv[1:] if "/" in a] or [
    (0, 0), (4, 1), (8, 1), (16, 1), (4, 2), (8, 2), (16, 2), (4, 4), (8, 4), (16, 4), (4, 8), (8, 8), (4, 16)]


def time_shape(B, Hq, Hkv, D, kv, layers=8, iters=20):
    dev = torch.device("cuda:0")
    bs = 16
    nblk = (kv + bs - 1) // bs
    nb = B * nblk + 8
    caches = []
    for _ in range(layers):
        kc = torch.empty(nb, Hkv, D // 8, bs, 8, dtype=torch.bfloat16, device=dev).uniform_(-0.09, 0.09)
        vc = torch.empty(nb, Hkv, D, bs, dtype=torch.bfloat16, device=dev).uniform_(-0.09, 0.09)
        caches.append((kc, vc))
    q = torch.empty(B, Hq, D, dtype=torch.bfloat16, device=dev).uniform_(-0.09, 0.09)
    bt = torch.randperm(nb - 8, device=dev)[:B * nblk].to(torch.int32).reshape(B, nblk)
    lens = torch.full((B,), kv, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    res = []
    ref = None
    for w, s in COMBOS:
        os.environ["PE_DECODE_WAVES"] = str(w)
        os.environ["PE_DECODE_SPLITS"] = str(s)
        ops.reload_env()

        def run(i):
            kc, vc = caches[i % layers]
            ops.paged_attention_v1(out, q, kc, vc, Hkv, D ** -0.5, bt, lens, bs, kv, None, "auto", 1.0, 1.0)
        try:
            for i in range(layers):
                run(i)
            torch.cuda.synchronize()
            # the launches of `layers` distinct caches captured into one graph: replays time the
            # GPU alone (eager launches from Python cost about as much as a small launch runs)
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
            res.append(st.elapsed_time(en) * 1e3 / (iters * layers))
            run(0)
            o = out.float().clone()
            if ref is None:
                ref = o
            elif not torch.allclose(o, ref, atol=2e-3, rtol=2e-2):
                res[-1] = float("nan")
        except RuntimeError:
            res.append(float("inf"))
    return res


def main():
    print("B Hq Hkv D kv | " + " ".join(f"{w}w/{s}s".rjust(7) for w, s in COMBOS))
    for shp in SHAPES:
        r = time_shape(*shp)
        best = min(range(min(1, len(COMBOS) - 1), len(COMBOS)), key=lambda i: r[i] if r[i] == r[i] else 1e9)
        print(" ".join(str(x) for x in shp) + " | " + " ".join(f"{t:7.2f}" for t in r)
              + f"   best {COMBOS[best][0]}w/{COMBOS[best][1]}s", flush=True)


if __name__ == "__main__":
    main()
