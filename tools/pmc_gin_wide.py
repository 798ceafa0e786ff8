"""Summarise `rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d DIR -o g --
python tools/gin_roofline.py --iters 2 --warmup 1` for the fused 8-layer launch of gin_wide_kernel (the longest
dispatches) into a JSON with the derived shares (MI355X_MICROARCH.md, rocprofv3 PMC slots).

    python tools/pmc_gin_wide.py DIR out.json
"""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    folder, out = sys.argv[1:3]
    rows = {}
    for path in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if "gin_wide" not in r["Kernel_Name"] or "pack" in r["Kernel_Name"]:
                    continue
                d = rows.setdefault(r["Dispatch_Id"], dict(dur=int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), c={}))
                d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if not rows:
        raise SystemExit("no gin_wide dispatches found")
    longest = max(v["dur"] for v in rows.values())
    fused = [v for v in rows.values() if v["dur"] > 0.6 * longest]
    c = {k: sum(v["c"].get(k, 0.0) for v in fused) / len(fused) for k in fused[0]["c"]}
    dur = sum(v["dur"] for v in fused) / len(fused) / 1e3
    wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    src = open(os.path.join(ROOT, "gcc_amd", "csrc", "gin_wide.hip"), "rb").read()
    rec = dict(kernel="gin_wide2_kernel (default shape), fused 8-layer launch (the longest dispatches of tools/gin_roofline.py)",
               source_sha256=hashlib.sha256(src).hexdigest(), dispatches=len(fused), duration_us_under_profiler=dur, counters=c,
               derived=dict(parked_on_waitcnt_or_barrier=c.get("SQ_WAIT_ANY", 0) / wc, issue_stalled=c.get("SQ_WAIT_INST_ANY", 0) / wc,
                            issuing=c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                            lds_conflict_share_of_lds_cycles=c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0), 1.0),
                            mfma_busy_cycles_per_simd=c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0))
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
