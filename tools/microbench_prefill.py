"""Times the prefill attention kernel alone."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vllm_paged_compression_amd.ops as ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--lq", type=int, default=1024)
    ap.add_argument("--lkv", type=int, default=1024)
    ap.add_argument("--hq", type=int, default=32)
    ap.add_argument("--hkv", type=int, default=8)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dump", default=None, help="save the output tensor here (A/B comparisons)")
    ap.add_argument("--lib", default=None, help="alternative build of the library (timing experiments)")
    a = ap.parse_args()
    if a.lib:
        import vllm_paged_compression_amd._lib as _lib
        _lib.LIB_PATH = os.path.abspath(a.lib)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    q = torch.randn(a.batch * a.lq, a.hq, a.d, dtype=torch.bfloat16, device=dev)
    k = torch.randn(a.batch * a.lkv, a.hkv, a.d, dtype=torch.bfloat16, device=dev)
    v = torch.randn(a.batch * a.lkv, a.hkv, a.d, dtype=torch.bfloat16, device=dev)
    cu_q = torch.arange(0, a.batch + 1, dtype=torch.int32, device=dev) * a.lq
    cu_kv = torch.arange(0, a.batch + 1, dtype=torch.int32, device=dev) * a.lkv
    out = torch.empty_like(q)
    sc = a.d ** -0.5
    for _ in range(3):
        ops.prefill_causal(q, k, v, cu_q, cu_kv, a.lq, sc, out=out)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(a.iters):
        ops.prefill_causal(q, k, v, cu_q, cu_kv, a.lq, sc, out=out)
    en.record()
    torch.cuda.synchronize()
    us = st.elapsed_time(en) * 1e3 / a.iters
    pairs = sum(min(i + 1, a.lkv) for i in range(a.lq))
    flops = 4.0 * a.batch * a.hq * a.d * pairs
    if a.dump:
        torch.save(out.cpu(), a.dump)
    chk = out.view(torch.int16).to(torch.int64).sum().item()   # bit-level checksum for A/B runs
    print(f"prefill B={a.batch} Lq={a.lq} Lkv={a.lkv}: {us:.1f} us, {flops / us / 1e6:.1f} TFLOP/s, checksum {chk}", flush=True)


if __name__ == "__main__":
    main()
