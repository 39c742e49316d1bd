"""Runs the phase-stamped prefill build: average shader-clock cycles per tile and wave in each
phase of the tile loop."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vllm_paged_compression_amd._lib as _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "_exp", "libpe_pfphase.so")
import vllm_paged_compression_amd.ops as ops  # noqa: E402


def main():
    B, L, Hq, Hkv, D = (int(x) for x in (sys.argv[1:6] + [8, 1024, 32, 8, 128][len(sys.argv) - 1:]))
    dev = torch.device("cuda:0")
    q = torch.randn(B * L, Hq, D, dtype=torch.bfloat16, device=dev)
    k = torch.randn(B * L, Hkv, D, dtype=torch.bfloat16, device=dev)
    v = torch.randn(B * L, Hkv, D, dtype=torch.bfloat16, device=dev)
    cu = torch.arange(0, B + 1, dtype=torch.int32, device=dev) * L
    out = torch.empty_like(q)
    for _ in range(3):
        ops.prefill_causal(q, k, v, cu, cu, L, D ** -0.5, out=out)
    torch.cuda.synchronize()
    lib = _lib.load()
    lib.pe_debug_read_pf_phases.restype = ctypes.c_int
    n = min(Hq * B * (L // 128), 8192) * 4
    buf = np.zeros(n * 8, dtype=np.uint64)
    assert lib.pe_debug_read_pf_phases(ctypes.c_void_p(buf.ctypes.data), ctypes.c_int(n * 8)) == 0
    ph = buf.reshape(n, 8).astype(np.float64)
    tiles = ph[:, 7] * 0 + 1
    computed = ph[:, 6].sum()
    loops = None
    names = ["QK (K reads + 32 MFMA)", "softmax", "P.V (V reads + 36 MFMA)", "stage issue", "wait staged tile",
             "barrier"]
    print(f"{int(computed)} computed wave-tiles")
    for i, nm in enumerate(names[:3]):
        print(f"  {nm:28s} {ph[:, i].sum() / computed:8.0f} cycles per computed tile")
    # loop-level phases are per loop iteration (a wave may skip the compute of a masked tile)
    iters = ph[:, 7] > 0
    total_iter_cycles = ph[:, 7].sum()
    print(f"  loop iteration total / computed tile {total_iter_cycles / computed:8.0f}")
    for i, nm in zip((3, 4, 5), names[3:]):
        print(f"  {nm:28s} {ph[:, i].sum() / computed:8.0f} cycles per computed tile")


if __name__ == "__main__":
    main()
