"""Host logic of bench.py that needs no GPU: the self-launcher for --gpus N (the driver calls `python bench.py --gpus N`
without a launcher), the workload label derived from the flags, whole-chunk timing windows, and the refusal of a stale
PMC traffic file."""
import json
import os
import subprocess
import sys
import textwrap

import bench


def test_launcher_command_starts_n_ranks_on_loopback(tmp_path):
    script = tmp_path / "ranks.py"
    script.write_text(textwrap.dedent("""
        import os, sys, torch, torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.tensor([float(dist.get_rank() + 1)])
        dist.all_reduce(t)
        if dist.get_rank() == 0:
            print("RESULT", int(t.item()), os.environ["WORLD_SIZE"], os.environ["MASTER_ADDR"], sys.argv[1:])
        dist.destroy_process_group()
    """))
    cmd = bench.launcher_command(2, ["--gpus", "2", "--steps", "4"], script=str(script))
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0]
    assert line == "RESULT 3 2 127.0.0.1 ['--gpus', '2', '--steps', '4']"


def test_workload_label_follows_the_flags():
    a = bench.parse_args([])
    assert (a.nodes, a.edges, a.mode) == (1_000_000, 10_000_000, "train")
    assert bench.workload_name(a, 1, 961441, 9938200).startswith("BASELINE configs[1]: MoCo K=16384")
    assert bench.workload_name(a, 8, 961441, 9938200).startswith("BASELINE configs[2]: ")
    b = bench.parse_args(["--mode", "sampler"])
    assert (b.nodes, b.edges) == (10_000_000, 200_000_000)
    assert bench.workload_name(b, 1, 9_960_000, 199_000_000).startswith("BASELINE configs[3]: sampler-only")
    c = bench.parse_args(["--nodes", "50000", "--edges", "500000"])
    name = bench.workload_name(c, 1, 49000, 490000)
    assert "BASELINE" not in name and "50,000-node/500,000-edge" in name


def test_stale_pmc_file_is_refused(tmp_path, monkeypatch):
    a = bench.parse_args([])
    f = tmp_path / "pmc.json"
    monkeypatch.setattr(bench, "PMC_FILE", str(f))
    assert bench.committed_pmc_traffic(a, 1, 2)[0] is None
    key = "1/2/bsz256/hops256"
    f.write_text(json.dumps(dict(source_sha256="not this build", workloads={key: dict(induce_kernel_hbm_bytes_per_launch=5.0)})))
    val, why = bench.committed_pmc_traffic(a, 1, 2)
    assert val is None and "stale" in why
    f.write_text(json.dumps(dict(source_sha256=bench.sampler_source_hash(), workloads={key: dict(induce_kernel_hbm_bytes_per_launch=5.0)})))
    assert bench.committed_pmc_traffic(a, 1, 2)[0] == 5.0
    assert bench.committed_pmc_traffic(a, 3, 4)[0] is None          # another graph: no entry
    # counters collected with the default hub-row settings say nothing about a run that scans every row
    b = bench.parse_args(["--hub-degree", "-1"])
    val, why = bench.committed_pmc_traffic(b, 1, 2)
    assert val is None and "hub" in why


def test_committed_pmc_file_matches_this_build():
    """profiles/pmc_sampler.json is keyed by the hash of the sampler sources: a commit that touches them without re-collecting
    the counters makes bench.py report ``traffic: null`` (by design) -- this test says so before the GPU box does."""
    rec = json.load(open(bench.PMC_FILE))
    assert rec["source_sha256"] == bench.sampler_source_hash(), "re-collect profiles/pmc_sampler.json (scripts/gpu/r6_call.sh <tag> pmc)"
    assert any(k.endswith("/steps10") for k in rec["workloads"]) and any(k.endswith("/steps16") for k in rec["workloads"])


def test_pmc_summary_adds_the_size_classes_up(tmp_path, monkeypatch):
    """tools/pmc_sampler.py: a call of gcc_sample_multi launches the walk and the induction once per size class (different
    template instances); bytes per call = the sum of the instances' per-dispatch averages, FETCH_SIZE x 2 for the induction."""
    import csv
    import importlib.util

    spec = importlib.util.spec_from_file_location("pmc_sampler", os.path.join(os.path.dirname(bench.__file__), "tools", "pmc_sampler.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = [("void (anonymous namespace)::induce_kernel<256>(int const*)", 100.0), ("void (anonymous namespace)::induce_kernel<256>(int const*)", 300.0),
            ("void (anonymous namespace)::induce_kernel<512>(int const*)", 1000.0), ("void (anonymous namespace)::induce_kernel<512>(int const*)", 3000.0),
            ("(anonymous namespace)::pack_kernel(int)", 50.0), ("at::native::something", 999.0)]
    for d, counter, scale in (("f", "FETCH_SIZE", 1.0), ("w", "WRITE_SIZE", 0.5)):
        os.makedirs(tmp_path / d / "x")
        with open(tmp_path / d / "x" / "p_counter_collection.csv", "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
            for name, v in rows:
                wr.writerow([name, counter, v * scale])
    fetch = mod.per_kernel(str(tmp_path / "f"), "FETCH_SIZE")
    assert fetch["induce_kernel"][0] == 200.0 + 2000.0 and fetch["pack_kernel"][0] == 50.0 and "something" not in str(fetch.keys())
    out = tmp_path / "pmc.json"
    monkeypatch.setattr(sys, "argv", ["pmc_sampler.py", str(tmp_path / "f"), str(tmp_path / "w"), "1/2/bsz256/hops256/steps16", str(out)])
    mod.main()
    rec = json.load(open(out))
    ent = rec["workloads"]["1/2/bsz256/hops256/steps16"]
    assert rec["source_sha256"] == bench.sampler_source_hash()
    assert ent["induce_kernel_hbm_bytes_per_launch"] == (2200.0 * 2.0 + 1100.0) * 1024.0
