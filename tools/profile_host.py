"""cProfile of one benchmark generation (host side): where the Python time goes."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    sys.argv = [sys.argv[0]] + sys.argv[1:]
    a = bench.parse()
    dev = torch.device("cuda", 0)
    import vllm_paged_compression_amd as pkg
    pkg.load_library()
    wl = bench.Workload(a, dev, 1, 0, None)
    wl.generate()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wl.generate()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host issue time {1e3 * (t1 - t0):.1f} ms, drain {1e3 * (t2 - t1):.1f} ms", flush=True)
    pr = cProfile.Profile()
    pr.enable()
    wl.generate()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
