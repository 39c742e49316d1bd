"""Runs the stamped prefill build and prints the per-workgroup timeline of one launch."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vllm_paged_compression_amd._lib as _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "_exp", "libpe_pfstamps.so")
import vllm_paged_compression_amd.ops as ops  # noqa: E402


def main():
    B, L, Hq, Hkv, D = 8, 1024, 32, 8, 128
    dev = torch.device("cuda:0")
    q = torch.randn(B * L, Hq, D, dtype=torch.bfloat16, device=dev)
    k = torch.randn(B * L, Hkv, D, dtype=torch.bfloat16, device=dev)
    v = torch.randn(B * L, Hkv, D, dtype=torch.bfloat16, device=dev)
    cu = torch.arange(0, B + 1, dtype=torch.int32, device=dev) * L
    out = torch.empty_like(q)
    for _ in range(5):
        ops.prefill_causal(q, k, v, cu, cu, L, D ** -0.5, out=out)
    torch.cuda.synchronize()
    lib = _lib.load()
    lib.pe_debug_read_pf_stamps.restype = ctypes.c_int
    n = Hq * B * (L // 128)
    buf = np.zeros(n * 6, dtype=np.uint64)
    rc = lib.pe_debug_read_pf_stamps(ctypes.c_void_p(buf.ctypes.data), ctypes.c_int(n * 6))
    assert rc == 0, rc
    st = buf.reshape(n, 6)
    t = st[:, :4].astype(np.int64)
    t0 = t[:, 0].min()
    t = (t - t0) * 0.01  # us
    tiles = st[:, 5].astype(np.int64)
    hw = st[:, 4]
    cu_id = ((hw & 0xffffffff) >> 8) & 0xf
    se_id = ((hw & 0xffffffff) >> 13) & 0x7
    sh_id = ((hw & 0xffffffff) >> 12) & 0x1
    xcc = (hw >> 32) & 0xf
    slot = xcc * 1000 + se_id * 100 + sh_id * 50 + cu_id

    def pct(x):
        return " ".join(f"{np.percentile(x, p):7.2f}" for p in (0, 10, 50, 90, 100))
    print("kernel span %.1f us, %d workgroups" % (t[:, 3].max(), n))
    print("percentiles: min p10 p50 p90 max")
    print("  start           ", pct(t[:, 0]))
    print("  startup (Q + first tile)", pct(t[:, 1] - t[:, 0]))
    print("  epilogue        ", pct(t[:, 3] - t[:, 2]))
    for nt in sorted(set(tiles.tolist())):
        sel = tiles == nt
        loop = t[sel, 2] - t[sel, 1]
        print(f"  tiles {nt:2d}: n={sel.sum():4d} loop {pct(loop)}  per tile {np.median(loop) / nt:5.2f} us, "
              f"start p50 {np.median(t[sel, 0]):6.1f}")
    # per-CU occupancy over time: busy workgroup-time / (2 slots * span)
    busy = (t[:, 3] - t[:, 0]).sum()
    ncu = len(set(slot.tolist()))
    print(f"distinct CUs seen {ncu}; workgroup-time / (2 * CUs * span) = {busy / (2 * ncu * t[:, 3].max()):.2f}")
    loop_t = (t[:, 2] - t[:, 1]).sum()
    print(f"share of workgroup time in the tile loop {loop_t / busy:.2f}, startup {(t[:, 1] - t[:, 0]).sum() / busy:.2f}, "
          f"epilogue {(t[:, 3] - t[:, 2]).sum() / busy:.2f}")
    last = np.sort(t[:, 3])[-512:]
    print("exit times of the last 512 workgroups: p0 %.1f p50 %.1f p100 %.1f" % (last[0], np.median(last), last[-1]))


if __name__ == "__main__":
    main()
