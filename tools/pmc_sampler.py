"""Summarise the separate `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes over tools/sampler_alone.py into
profiles/pmc_sampler.json (bench.py's roofline.traffic).  The file records the sha256 of the kernel source it was
collected from and the workload key; bench.py refuses it for any other build or workload.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_f -o f -- python tools/sampler_alone.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_w -o w -- python tools/sampler_alone.py
    python tools/pmc_sampler.py gpurun_out/pmc_f gpurun_out/pmc_w <V/E/bszB/hopsH> [profiles/pmc_sampler.json]

Units and correction (MI355X_MICROARCH.md, HBM section): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB of fabric
(L2 memory-side) requests, Infinity-Cache hits included; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide
(16 B per lane) coalesced reads at 64 bytes, so kernels whose traffic is such streams are doubled: induce_kernel
(dwordx4 row scans).  Other kernels and WRITE_SIZE are reported raw (uncalibrated widths).  "Per launch" = per call of
gcc_sample_multi: the two size classes of the walk / the induction are separate dispatches and are added up."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WIDE = ("induce_kernel",)


def per_kernel(folder, counter):
    acc = {}
    files = glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {folder}")
    for path in files:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                full = row["Kernel_Name"]
                name = next((k for k in ("rwr_walk_kernel", "induce_kernel", "pack_kernel", "prefix_a_kernel", "prefix_b_kernel", "records_kernel",
                                         "hub_write_kernel")
                             if k in full), None)
                if name is None:
                    continue
                s, n = acc.get((name, full), (0.0, 0))
                acc[(name, full)] = (s + float(row["Counter_Value"]), n + 1)
    # a call launches the walk and the induction once per size class (different template instances): per call = the sum of
    # the instances' per-dispatch averages
    out = {}
    for (name, full), (s, n) in acc.items():
        ps, pn = out.get(name, (0.0, 0))
        out[name] = (ps + s / n, max(pn, n))
    return out


def main():
    import bench

    fdir, wdir, key = sys.argv[1:4]
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles", "pmc_sampler.json")
    fetch, write = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    rec = json.load(open(out)) if os.path.exists(out) else {}
    sha = bench.sampler_source_hash()
    if rec.get("source_sha256") != sha:
        rec = {"source_sha256": sha, "workloads": {}}
    rec["units"] = "KiB per launch as rocprofv3 reports them; *_bytes fields are bytes per launch"
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(name, (0.0, 0))
        w, nw = write.get(name, (0.0, 0))
        wide = any(t in name for t in WIDE)
        kernels[name] = dict(FETCH_SIZE_KB_per_launch_raw=f, WRITE_SIZE_KB_per_launch_raw=w, launches=max(nf, nw),
                             fetch_correction=2.0 if wide else 1.0,
                             hbm_bytes_per_launch=(f * (2.0 if wide else 1.0) + w) * 1024.0)
    ind = [v for k, v in kernels.items() if "induce" in k]
    rec["workloads"][key] = dict(
        kernels=kernels, induce_kernel_hbm_bytes_per_launch=ind[0]["hbm_bytes_per_launch"] if ind else None,
        correction="FETCH_SIZE x2 for the dwordx4 row scans of induce_kernel (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE raw")
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec["workloads"][key], indent=1))


if __name__ == "__main__":
    main()
