"""Wall time of consecutive benchmark generations (each followed by a device sync)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    a = bench.parse()
    import vllm_paged_compression_amd as pkg
    pkg.load_library()
    wl = bench.Workload(a, torch.device("cuda", 0), 1, 0, None)
    import gc
    for i in range(6):
        if i == 2:
            wl.timing_on = True
            if os.environ.get("PE_GC_FREEZE"):
                gc.collect()
                gc.freeze()
        t0 = time.perf_counter()
        wl.generate()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"generation {i}: host {1e3 * (t1 - t0):.1f} ms, total {1e3 * (t2 - t0):.1f} ms", flush=True)


if __name__ == "__main__":
    main()
