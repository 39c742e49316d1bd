"""Decode and prefill attention per head size of the reference's op (vllm/attention/ops/paged_attn.py:37):
matrix-core kernel against the general kernel (PE_DECODE_FORCE_GENERIC / PE_PREFILL_FORCE_GENERIC),
us per launch, graph-replayed for decode.  Prints a markdown table."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vllm_paged_compression_amd.ops as ops  # noqa: E402

dev = torch.device("cuda:0")


def decode_us(D, B=64, Hq=32, Hkv=8, kv=776, layers=4, iters=10, bs=16):
    nblk = (kv + bs - 1) // bs
    nb = B * nblk + 8
    caches = [(torch.empty(nb, Hkv, D // 8, bs, 8, dtype=torch.bfloat16, device=dev).uniform_(-0.09, 0.09),
               torch.empty(nb, Hkv, D, bs, dtype=torch.bfloat16, device=dev).uniform_(-0.09, 0.09))
              for _ in range(layers)]
    q = torch.empty(B, Hq, D, dtype=torch.bfloat16, device=dev).uniform_(-0.09, 0.09)
    bt = torch.randperm(nb - 8, device=dev)[:B * nblk].to(torch.int32).reshape(B, nblk)
    lens = torch.full((B,), kv, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)

    def run(i):
        kc, vc = caches[i % layers]
        ops.paged_attention_v1(out, q, kc, vc, Hkv, D ** -0.5, bt, lens, bs, kv, None, "auto", 1.0, 1.0)
    for i in range(layers):
        run(i)
    torch.cuda.synchronize()
    side, graph = torch.cuda.Stream(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        for i in range(layers):
            run(i)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for i in range(layers):
                run(i)
    graph.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        graph.replay()
    e.record()
    torch.cuda.synchronize()
    nbytes = B * (2 * kv * Hkv * D * 2 + nblk * 4) + 2 * B * Hq * D * 2
    return s.elapsed_time(e) * 1e3 / (iters * layers), nbytes


def prefill_us(D, n=8, L=1024, Hq=32, Hkv=8, iters=5):
    q = torch.empty(n * L, Hq, D, dtype=torch.bfloat16, device=dev).uniform_(-1, 1)
    k = torch.empty(n * L, Hkv, D, dtype=torch.bfloat16, device=dev).uniform_(-1, 1)
    v = torch.empty(n * L, Hkv, D, dtype=torch.bfloat16, device=dev).uniform_(-1, 1)
    cu = torch.arange(0, (n + 1) * L, L, dtype=torch.int32, device=dev)
    ops.prefill_causal(q, k, v, cu, cu, L, D ** -0.5)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.prefill_causal(q, k, v, cu, cu, L, D ** -0.5)
    e.record()
    torch.cuda.synchronize()
    flops = 4.0 * Hq * D * (L * (L + 1) // 2) * n
    return s.elapsed_time(e) * 1e3 / iters, flops


def main():
    print("| head size | decode 64 x (32q/8kv) kv 776: matrix core us (TB/s) | general us | "
          "prefill 8 x 1024: matrix core us (TFLOP/s) | general us |")
    print("|---|---|---|---|---|")
    for D in (32, 64, 80, 96, 112, 120, 128, 192, 256):
        row = [str(D)]
        for kind in ("decode", "prefill"):
            var = "PE_DECODE_FORCE_GENERIC" if kind == "decode" else "PE_PREFILL_FORCE_GENERIC"
            os.environ.pop(var, None)
            ops.reload_env()
            if kind == "decode":
                t, nbytes = decode_us(D)
                row.append(f"{t:.1f} ({nbytes / t / 1e6:.2f})")
            else:
                t, flops = prefill_us(D)
                row.append(f"{t:.1f} ({flops / t / 1e6:.0f})")
            os.environ[var] = "1"
            ops.reload_env()
            t2, _ = decode_us(D, iters=2) if kind == "decode" else prefill_us(D, iters=1)
            row.append(f"{t2:.0f}")
            os.environ.pop(var, None)
            ops.reload_env()
        print("| " + " | ".join(row) + " |", flush=True)
    # decode at the other page sizes of the reference's study (scripts/diff_page_sizes.sh:16)
    print()
    print("| head size | decode at 8-token blocks: matrix core us (TB/s) | general us | "
          "at 32-token blocks: matrix core us (TB/s) | general us |")
    print("|---|---|---|---|---|")
    for D in (32, 64, 96, 128, 192, 256):
        row = [str(D)]
        for bs in (8, 32):
            os.environ.pop("PE_DECODE_FORCE_GENERIC", None)
            ops.reload_env()
            t, nbytes = decode_us(D, bs=bs)
            row.append(f"{t:.1f} ({nbytes / t / 1e6:.2f})")
            os.environ["PE_DECODE_FORCE_GENERIC"] = "1"
            ops.reload_env()
            t2, _ = decode_us(D, iters=2, bs=bs)
            row.append(f"{t2:.0f}")
            os.environ.pop("PE_DECODE_FORCE_GENERIC", None)
            ops.reload_env()
        print("| " + " | ".join(row) + " |", flush=True)


if __name__ == "__main__":
    main()
