"""Runs the stamped decode build and prints the per-workgroup timeline of one launch."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vllm_paged_compression_amd._lib as _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "_exp", "libpe_stamps.so")
import vllm_paged_compression_amd.ops as ops  # noqa: E402


def main():
    B, kv, Hq, Hkv, D, bs = 64, 776, 32, 8, 128, 16
    dev = torch.device("cuda:0")
    nblk = (kv + bs - 1) // bs
    nb = B * nblk + 8
    caches = []
    for _ in range(8):
        kc = torch.empty(nb, Hkv, D // 8, bs, 8, dtype=torch.bfloat16, device=dev).uniform_(-0.09, 0.09)
        vc = torch.empty(nb, Hkv, D, bs, dtype=torch.bfloat16, device=dev).uniform_(-0.09, 0.09)
        caches.append((kc, vc))
    q = torch.empty(B, Hq, D, dtype=torch.bfloat16, device=dev).uniform_(-0.09, 0.09)
    bt = torch.randperm(nb - 8, device=dev)[:B * nblk].to(torch.int32).reshape(B, nblk)
    lens = torch.full((B,), kv, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    for i in range(24):
        kc, vc = caches[i % 8]
        ops.paged_attention_v1(out, q, kc, vc, Hkv, D ** -0.5, bt, lens, bs, kv, None, "auto", 1.0, 1.0)
    torch.cuda.synchronize()
    lib = _lib.load()
    n = B * Hkv
    buf = np.zeros(n * 4, dtype=np.uint64)
    rc = lib.pe_debug_read_stamps(ctypes.c_void_p(buf.ctypes.data), ctypes.c_int(n * 4))
    assert rc == 0, rc
    st = buf.reshape(n, 4).astype(np.int64)
    t0 = st[:, 0].min()
    st = (st - t0) * 0.01  # us
    def pct(x):
        return " ".join(f"{np.percentile(x, p):6.2f}" for p in (0, 10, 50, 90, 100))
    print("percentiles (us since first workgroup start): min p10 p50 p90 max")
    print("  start      ", pct(st[:, 0]))
    print("  first tile ", pct(st[:, 1]))
    print("  loop done  ", pct(st[:, 2]))
    print("  exit       ", pct(st[:, 3]))
    print("  first-tile latency", pct(st[:, 1] - st[:, 0]))
    print("  loop duration     ", pct(st[:, 2] - st[:, 1]))
    print("  epilogue          ", pct(st[:, 3] - st[:, 2]))
    wg = np.arange(n)
    print("loop-done by workgroup id % 8 (XCD of the round-robin dispatch): median / max")
    for x in range(8):
        sel = st[wg % 8 == x]
        print(f"  {x}: {np.median(sel[:, 2]):6.2f} {sel[:, 2].max():6.2f}   first tile {np.median(sel[:, 1]):6.2f}")
    print("loop-done by dispatch order (64 consecutive workgroup ids per row): median")
    print("  ", " ".join(f"{np.median(st[i:i + 64, 2]):5.1f}" for i in range(0, n, 64)))


if __name__ == "__main__":
    main()
