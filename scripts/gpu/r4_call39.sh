#!/bin/bash
# Round 4, call 39: memory-side traffic of the eigensolver kernels (one 16-view call, tools/posemb_phases.py): FETCH_SIZE / WRITE_SIZE per kernel
set -u
O=gpurun_out/r4c39
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && rm -rf /tmp/pp_$c && (timeout 600 rocprofv3 --output-format csv --pmc $c --kernel-trace -d /tmp/pp_$c -o p -- python $GRAFT_REPO_ROOT/tools/posemb_phases.py 2>&1 | tail -1) > /dev/null; cd $GRAFT_REPO_ROOT
done
python - <<'PY' | tee $O/posemb_traffic.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0.0, 0, 0.0])
for c, slot in (("FETCH_SIZE", 0), ("WRITE_SIZE", 1)):
    for f in glob.glob(f"/tmp/pp_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != c: continue
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if not n.startswith("posemb"): continue
            acc[n][slot] += float(r["Counter_Value"])
            if slot == 0: acc[n][2] += 1
for f in glob.glob("/tmp/pp_FETCH_SIZE/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if n in acc: acc[n][3] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("kernel, dispatches, FETCH_SIZE MB (raw KiB counters / 1024), WRITE_SIZE MB, total us, fetch GB/s over the kernel's own duration")
for n, (f, w, k, us) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{n:48s} {k:3d}  fetch {f / 1024:9.1f} MB  write {w / 1024:9.1f} MB  {us:10.1f} us  {f / 1024 / 1e3 / max(us, 1e-9) * 1e6:8.1f} GB/s")
PY
