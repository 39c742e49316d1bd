"""GPU idle-gap report from a rocprofv3 --kernel-trace CSV: busy time, idle time, and the idle
time grouped by the kernel that follows the gap."""
import collections
import csv
import sys


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = rows[skip:]
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    gaps = collections.defaultdict(lambda: [0, 0])
    big = 0
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        g = s1 - e0
        if g > 0:
            key = n1.split("(")[0][:60]
            gaps[key][0] += g
            gaps[key][1] += 1
            if g > 20000:
                big += g
    print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  idle {(span - busy) / 1e6:.2f} ms"
          f"  (gaps > 20 us: {big / 1e6:.2f} ms)")
    for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"  idle before {k:62s} {g / 1e6:8.2f} ms in {n:6d} gaps ({g / max(n, 1) / 1e3:.1f} us avg)")


if __name__ == "__main__":
    main()
