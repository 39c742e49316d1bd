"""Where the time between kernels goes: reads a rocprofv3 kernel-trace CSV and prints, over the last
``--frac`` of the trace (the timed generations), GPU-busy time against wall time and the idle gaps
grouped by (kernel before, kernel after).

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 bench.py --config c1 ...
    python3 tools/trace_gaps.py out/*/t_kernel_trace.csv
"""
import argparse
import collections
import csv
import re


def short(name: str) -> str:
    m = re.search(r"(paged_decode_mfma|paged_decode_merge|prefill_mfma|reshape_and_cache|table_state_advance|"
                  r"evict_select\w*|copyBuffer|token_\w+|rope_cache|block_score\w*|zero_slots\w*)", name)
    return m.group(1) if m else name[:40]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--frac", type=float, default=0.5)
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    rows = rows[int(len(rows) * (1 - a.frac)):]
    wall = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    print(f"kernels {len(rows)}  wall {wall / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  idle {(wall - busy) / 1e6:.3f} ms")
    gaps = collections.defaultdict(list)
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        gaps[(n0, n1)].append(max(0, s1 - e0))
    print(f"{'before -> after':58s} {'n':>6s} {'total ms':>9s} {'avg us':>8s} {'max us':>8s}")
    for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k[0] + ' -> ' + k[1]:58s} {len(v):6d} {sum(v) / 1e6:9.3f} {sum(v) / len(v) / 1e3:8.2f} {max(v) / 1e3:8.2f}")
    dur = collections.defaultdict(list)
    for s, e, n in rows:
        dur[n].append(e - s)
    for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {n:40s} {len(v):6d} x {sum(v) / len(v) / 1e3:8.2f} us = {sum(v) / 1e6:8.3f} ms")


if __name__ == "__main__":
    main()
