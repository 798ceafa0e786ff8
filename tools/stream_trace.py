"""The launch sequence of ONE training step from a rocprofv3 kernel trace of tools/graph_probe.py (training stream alone,
fixed batch): per launch the start offset inside the step, duration and the idle gap before it.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python tools/graph_probe.py --steps 40
    python tools/stream_trace.py /tmp/tr [step index, default: the median-duration eager step]
"""
import csv
import glob
import os
import sys

root = sys.argv[1]
path = [p for p in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)][0]
rows = list(csv.DictReader(open(path)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
first = [i for i, e in enumerate(ev) if "gin_feat_kernel" in e[2]]
steps = [(first[i], first[i + 1]) for i in range(len(first) - 1)]
# eager steps only (the hipGraph replays come after; same kernels, so just take the first 50 steps)
steps = steps[5:45]
dur = sorted((ev[b - 1][1] - ev[a][0], a, b) for a, b in steps)
d, a, b = dur[len(dur) // 2] if len(sys.argv) < 3 else [(ev[y - 1][1] - ev[x][0], x, y) for x, y in steps][int(sys.argv[2])]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:44]


print(f"median step: {d / 1e3:.1f} us from the first launch's start to the last one's end, {b - a} launches; "
      f"step period {(ev[b][0] - ev[a][0]) / 1e3:.1f} us")
t0, prev_end, busy, gaps = ev[a][0], ev[a][0], 0, 0
for s, e, n in ev[a:b]:
    gap = s - prev_end
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f}  gap {gap / 1e3:6.1f}  {short(n)}")
    busy += e - s
    gaps += max(gap, 0)
    prev_end = max(prev_end, e)
print(f"busy {busy / 1e3:.1f} us, gaps {gaps / 1e3:.1f} us; gap to the next step's first launch {(ev[b][0] - prev_end) / 1e3:.1f} us")
agg = {}
for s, e, n in ev[a:b]:
    k = short(n)
    c = agg.setdefault(k, [0, 0])
    c[0] += 1
    c[1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:44s} x{c:3d}  {t / 1e3:7.1f} us")
