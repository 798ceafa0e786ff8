"""Where a step's time goes INSIDE the benchmarked pipeline (producers running): from a rocprofv3 kernel trace of bench.py, per
training step (gin_feat_kernel start to the next one's) the busy time of the training stream's kernels, the idle gaps between
them, and the longest gap -- next to what the same kernels take with the GPU to themselves (tools/stream_trace.py).

    rocprofv3 --kernel-trace --output-format csv -d /tmp/trb -o t -- python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-parity
    python tools/bench_step_trace.py /tmp/trb
"""
import csv
import glob
import os
import sys

import numpy as np

root = sys.argv[1]
path = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(path)))
TRAIN = ("gin_", "nce_", "adam_kernel", "gradnorm_kernel", "queue_enqueue_kernel", "step_scalars")
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows
             if any(t in r["Kernel_Name"] for t in TRAIN)), key=lambda e: e[0])
first = [i for i, e in enumerate(ev) if "gin_feat_kernel" in e[2]]
steps = [(first[i], first[i + 1]) for i in range(len(first) - 1)]
steps = steps[len(steps) // 3:]                      # the later steps: graph replays, steady pipeline
per, busy, gaps, worst, names = [], [], [], [], {}
for a, b in steps:
    t0, t1 = ev[a][0], ev[b][0]
    per.append((t1 - t0) / 1e3)
    bz, prev_end, g, w = 0, ev[a][0], 0, (0, "")
    for s, e, n in ev[a:b]:
        bz += e - s
        gap = max(s - prev_end, 0)
        g += gap
        if gap > w[0]:
            w = (gap, n)
        prev_end = max(prev_end, e)
        short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
        names.setdefault(short, []).append((e - s) / 1e3)
    tail = max(t1 - prev_end, 0)                     # from the step's last kernel to the next step's first
    busy.append(bz / 1e3)
    gaps.append((g + tail) / 1e3)
    worst.append((w[0] / 1e3, w[1].split("(")[0][-30:], tail / 1e3))
per, busy, gaps = np.array(per), np.array(busy), np.array(gaps)
print(f"{len(per)} steps: period {per.mean():.1f} us (median {np.median(per):.1f}, p90 {np.percentile(per, 90):.1f}); training-stream kernels busy "
      f"{busy.mean():.1f} us; idle between them {gaps.mean():.1f} us of which after the step's last kernel {np.mean([w[2] for w in worst]):.1f} us")
print("longest in-step gaps (us, before kernel):", sorted(((round(w[0], 1), w[1]) for w in worst), reverse=True)[:5])
print("per kernel, mean us per launch inside the pipeline (x launches per step):")
for n, v in sorted(names.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n:42s} x{len(v) / len(per):5.1f}  {np.mean(v):7.1f}")
