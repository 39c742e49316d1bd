"""Turns the raw output of tools/collect_profiles.sh (gpurun_out/prof/) into the committed
summaries under profiles/:

  <tag>_bench_n1.json            the bench line of the un-profiled run
  <tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same bench command
  <tag>_decode_pmc.json          HBM traffic per decode launch from the FETCH_SIZE / WRITE_SIZE
                                 passes (separate runs), corrected as the gfx950 notes prescribe:
                                 FETCH_SIZE x2 for wide coalesced reads, WRITE_SIZE as reported
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_avg(path, pattern):
    vals = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if pattern in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in vals.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "round1"
    src = os.path.join(ROOT, "gpurun_out", "prof")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    bench = json.loads(open(os.path.join(src, "bench_n1.json")).read().strip().splitlines()[-1])
    with open(os.path.join(dst, f"{tag}_bench_n1.json"), "w") as f:
        f.write(json.dumps(bench) + "\n")
    shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"),
                os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
    kernel = "paged_decode_mfma_kernel"
    fetch, nf = counter_avg(os.path.join(src, "pmc_fetch", "fetch_counter_collection.csv"), kernel)["FETCH_SIZE"]
    write, nw = counter_avg(os.path.join(src, "pmc_write", "write_counter_collection.csv"), kernel)["WRITE_SIZE"]
    hbm = int(fetch * 1024 * 2 + write * 1024)
    algo = bench["roofline"]["algorithmic_bytes_per_launch"]
    stats_avg = None
    with open(os.path.join(src, "stats", "bench_kernel_stats.csv")) as f:
        for r in csv.DictReader(f):
            if kernel in r["Name"]:
                stats_avg = float(r["AverageNs"]) / 1e3
    cfg = bench["config"]["workload"]
    out = {
        "kernel": kernel,
        "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py --steps 1 --warmup 0 "
                   "--no-cpu-baseline --alt-evict 0 (separate passes; average over every decode launch of the run)",
        "workload": cfg,
        "launches": {"fetch_pass": nf, "write_pass": nw},
        "FETCH_SIZE_kb_raw": fetch, "WRITE_SIZE_kb_raw": write,
        "correction": "FETCH_SIZE x2 (gfx950 tallies the 128-B requests of wide coalesced reads as 64 B); "
                      "WRITE_SIZE as reported",
        "hbm_bytes_per_launch": hbm,
        "algorithmic_bytes_per_launch": algo,
        "traffic_over_algorithmic": round(hbm / algo, 4),
        "rocprof_avg_launch_us": stats_avg,
        "bench_avg_launch_us": bench["roofline"]["avg_launch_us"],
    }
    with open(os.path.join(dst, f"{tag}_decode_pmc.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
