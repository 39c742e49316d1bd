"""Average the per-dispatch counter values of one kernel from rocprofv3 counter_collection CSVs."""
import collections
import csv
import sys


def main():
    pattern = sys.argv[1]
    for path in sys.argv[2:]:
        acc = collections.defaultdict(list)
        with open(path) as f:
            for r in csv.DictReader(f):
                if pattern in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(f"{k:32s} {sum(v) / len(v):16.1f}  (n={len(v)})")


if __name__ == "__main__":
    main()
