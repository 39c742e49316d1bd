"""Decode step of all layers: eager launches vs one HIP-graph replay (wall time per step)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vllm_paged_compression_amd.ops as ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--kv", type=int, default=776)
    ap.add_argument("--hq", type=int, default=32)
    ap.add_argument("--hkv", type=int, default=8)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    D, bs = 128, 16
    B, M = a.batch, a.kv // bs + 2
    nb = B * M + 1
    kvs = [torch.zeros(2, nb, bs * a.hkv * D, dtype=torch.bfloat16, device=dev).uniform_(-0.1, 0.1)
           for _ in range(a.layers)]
    q = torch.zeros(B, a.hq * D, dtype=torch.bfloat16, device=dev).uniform_(-0.1, 0.1)
    k = torch.zeros(B, a.hkv * D, dtype=torch.bfloat16, device=dev).uniform_(-0.1, 0.1)
    v = k.clone()
    out = torch.zeros_like(q)
    bt = torch.randperm(nb - 1, device=dev)[:B * M].to(torch.int32).reshape(B, M)
    lens = torch.full((B,), a.kv, dtype=torch.int32, device=dev)
    slots = bt[:, (a.kv - 1) // bs].to(torch.int64) * bs + (a.kv - 1) % bs

    def step():
        launcher = ops.DecodeLayerLauncher(q, k, v, kvs[0], a.hq, a.hkv, D, slots, None, bt, lens,
                                           a.kv + bs, D ** -0.5)
        for li in range(a.layers):
            launcher.run(q, k, v, kvs[li], out)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / a.iters
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / a.iters
    print(f"B={B} kv={a.kv} Hq={a.hq} Hkv={a.hkv} layers={a.layers}: eager {eager*1e6:.0f} us/step "
          f"({eager*1e6/a.layers:.1f} us/layer), graph replay {graph*1e6:.0f} us/step "
          f"({graph*1e6/a.layers:.1f} us/layer)", flush=True)


if __name__ == "__main__":
    main()
