"""Host-side time of the phases of one benchmark generation (vectorised scheduler path)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    a = bench.parse()
    dev = torch.device("cuda", 0)
    import vllm_paged_compression_amd as pkg
    pkg.load_library()
    wl = bench.Workload(a, dev, 1, 0, None)
    wl.generate()
    torch.cuda.synchronize()
    marks = []
    orig_layers = wl._layers

    def layers(*args, **kw):
        t0 = time.perf_counter()
        orig_layers(*args, **kw)
        marks.append(("layers", t0, time.perf_counter()))

    wl._layers = layers
    t0 = time.perf_counter()
    wl.generate()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"generate host {1e3 * (t1 - t0):.1f} ms, drain {1e3 * (t2 - t1):.1f} ms")
    prev = t0
    gaps = []
    for _, s, e in marks:
        gaps.append((s - prev, e - s))
        prev = e
    print("before first step: %.2f ms" % (1e3 * gaps[0][0]))
    pre = gaps[:8]
    dec = gaps[8:]
    print("prefill steps: host between steps avg %.2f ms, layers avg %.2f ms" % (
        1e3 * sum(g for g, _ in pre[1:]) / max(len(pre) - 1, 1), 1e3 * sum(l for _, l in pre) / len(pre)))
    print("decode steps: host between steps avg %.3f ms (max %.2f), layers avg %.3f ms" % (
        1e3 * sum(g for g, _ in dec) / len(dec), 1e3 * max(g for g, _ in dec), 1e3 * sum(l for _, l in dec) / len(dec)))
    print("after last step: %.2f ms" % (1e3 * (t1 - prev)))


if __name__ == "__main__":
    main()
