"""Sweep of the decode kernel's KV split count over (batch, KV heads, kv length): prints the time
per launch for each forced split count (one process per point, PE_DECODE_SPLITS is read once)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(1, 8, 776), (1, 8, 4104), (1, 8, 16400), (4, 8, 4104), (8, 8, 776), (8, 8, 4104), (16, 8, 776),
          (64, 1, 776), (64, 1, 4104), (64, 2, 776), (64, 4, 776), (8, 1, 4104), (64, 8, 776)]
SPLITS = [0, 1, 2, 4, 8, 16, 32]   # 0 = the library's own choice


def main():
    print("batch hkv kv | " + " ".join(f"s={s:<5d}" for s in SPLITS))
    for b, hkv, kv in SHAPES:
        row = []
        for s in SPLITS:
            env = dict(os.environ, PE_DECODE_SPLITS=str(s))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "microbench_decode.py"),
                                  "--batch", str(b), "--hkv", str(hkv), "--hq", str(4 * hkv), "--kv", str(kv),
                                  "--iters", "10", "--layers", "16"], env=env, capture_output=True, text=True,
                                 timeout=120)
            try:
                row.append(float(out.stdout.split(": ")[1].split(" us")[0]))
            except Exception:
                row.append(float("nan"))
        best = min(range(len(SPLITS)), key=lambda i: row[i])
        print(f"{b:5d} {hkv:3d} {kv:5d} | " + " ".join(f"{t:7.2f}" for t in row) + f"   best s={SPLITS[best]}",
              flush=True)


if __name__ == "__main__":
    main()
