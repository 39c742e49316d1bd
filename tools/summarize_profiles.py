"""Turns the raw output of tools/collect_profiles.sh (gpurun_out/prof/) into the committed
summaries under profiles/:

  <tag>_bench_<cfg>.json            the bench line of the un-profiled run of that configuration
  <tag>_bench_<cfg>_kernel_stats.csv  rocprofv3 --kernel-trace --stats of the same bench command
  <tag>_decode_pmc_<cfg>.json       HBM traffic per decode launch from the FETCH_SIZE / WRITE_SIZE
                                    passes (separate runs), corrected as the gfx950 notes
                                    prescribe: FETCH_SIZE x2 for wide coalesced reads, WRITE_SIZE
                                    as reported; carries the hash of the decode kernel's sources
                                    so that bench.py only reuses the ratio for the same code
  <tag>_prefill_pmc.json            SQ counters of the prefill kernel on the two tracked shapes:
                                    matrix-pipe busy fraction, issue / stall / parked split, LDS
                                    bank conflicts, the clock held under load
"""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def read_counters(path):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r"(\S+)\s+([\d.]+)\s+\(n=(\d+)\)", line)
        if m:
            out[m.group(1)] = (float(m.group(2)), int(m.group(3)))
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "round3"
    src = os.path.join(ROOT, "gpurun_out", "prof")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    from bench import kernel_source_hash
    kernel = "paged_decode_mfma_kernel"
    for cfg in ("c1", "c2", "c3", "c4", "c5"):
        bpath = os.path.join(src, f"bench_{cfg}.json")
        if not os.path.exists(bpath) or not open(bpath).read().strip():
            continue
        bench = json.loads(open(bpath).read().strip().splitlines()[-1])
        with open(os.path.join(dst, f"{tag}_bench_{cfg}.json"), "w") as f:
            f.write(json.dumps(bench) + "\n")
        stats = os.path.join(src, f"stats_{cfg}", "bench_kernel_stats.csv")
        stats_avg = {}
        if os.path.exists(stats):
            shutil.copy(stats, os.path.join(dst, f"{tag}_bench_{cfg}_kernel_stats.csv"))
            with open(stats) as f:
                for r in csv.DictReader(f):
                    for k in (kernel, "prefill_mfma_kernel"):
                        if k in r["Name"]:
                            stats_avg[k] = float(r["AverageNs"]) / 1e3
        fetch = read_counters(os.path.join(src, f"pmc_fetch_{cfg}.txt")).get("FETCH_SIZE")
        write = read_counters(os.path.join(src, f"pmc_write_{cfg}.txt")).get("WRITE_SIZE")
        if fetch and write:
            hbm = int(fetch[0] * 1024 * 2 + write[0] * 1024)
            algo = bench["roofline"]["algorithmic_bytes_per_launch"]
            out = {
                "kernel": kernel,
                "command": f"rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py --config {cfg} "
                           "--steps 1 --warmup 0 --no-cpu-baseline --alt-evict 0 (separate passes; "
                           "average over every decode launch of the run)",
                "workload": bench["config"]["workload"],
                "kernel_source_sha256": kernel_source_hash(),
                "launches": {"fetch_pass": fetch[1], "write_pass": write[1]},
                "FETCH_SIZE_kb_raw": fetch[0], "WRITE_SIZE_kb_raw": write[0],
                "correction": "FETCH_SIZE x2 (gfx950 tallies the 128-B requests of wide coalesced reads "
                              "as 64 B); WRITE_SIZE as reported",
                "hbm_bytes_per_launch": hbm,
                "algorithmic_bytes_per_launch": algo,
                "traffic_over_algorithmic": round(hbm / algo, 4),
                "rocprof_avg_launch_us": stats_avg.get(kernel),
                "bench_avg_launch_us": bench["roofline"]["avg_launch_us"],
            }
            with open(os.path.join(dst, f"{tag}_decode_pmc_{cfg}.json"), "w") as f:
                json.dump(out, f, indent=1)
                f.write("\n")
            print(cfg, json.dumps(out))
    # prefill counters
    pf = {}
    for shape in ("b8_l1024", "b4_l8192"):
        c = read_counters(os.path.join(src, f"pmc_pf_{shape}.txt"))
        tpath = os.path.join(src, f"prefill_{shape}.txt")
        if not c or not os.path.exists(tpath):
            continue
        m = re.search(r"([\d.]+) us, ([\d.]+) TFLOP/s", open(tpath).read())
        us, tf = float(m.group(1)), float(m.group(2))
        g = lambda k: c[k][0] if k in c else None  # noqa: E731
        cycles = g("GRBM_GUI_ACTIVE") / 8.0 if g("GRBM_GUI_ACTIVE") else None   # summed over 8 XCDs
        simds = 256 * 4
        pf[shape] = {
            "command": f"tools/microbench_prefill.py --batch {shape[1]} --lq/--lkv {shape.split('_l')[1]} "
                       "(gaussian data, 32 query / 8 KV heads x 128)",
            "us_per_launch": us, "tflops": tf, "frac_of_2500_tflops": round(tf / 2500.0, 4),
            "counters_per_launch": {k: v[0] for k, v in c.items()},
            "shader_cycles_per_launch": cycles,
            "clock_ghz_under_pmc": round(cycles / us / 1e3, 3) if cycles else None,
            "mfma_pipe_busy_frac": round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (cycles * simds), 4) if cycles else None,
            "wave_cycles_split": {
                "issuing": round(g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), 4),
                "stalled_at_issue": round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 4),
                "parked_waitcnt_or_barrier": round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 4)},
            "lds_bank_conflict_cycles": g("SQ_LDS_BANK_CONFLICT"),
        }
    if pf:
        with open(os.path.join(dst, f"{tag}_prefill_pmc.json"), "w") as f:
            from bench import PREFILL_SOURCES
            json.dump({"kernel": "prefill_mfma_kernel", "shapes": pf,
                       "kernel_source_sha256": kernel_source_hash(PREFILL_SOURCES),
                       "notes": "mfma_pipe_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (shader cycles x 1024 "
                                "SIMDs), shader cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs); "
                                "the timed run is a separate un-profiled one"}, f, indent=1)
            f.write("\n")
        print(json.dumps(pf, indent=1))


if __name__ == "__main__":
    main()
