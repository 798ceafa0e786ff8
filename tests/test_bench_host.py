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
    assert rec["source_sha256"] == bench.sampler_source_hash(), "re-collect profiles/pmc_sampler.json (scripts/gpu/r4_call27.sh)"
    assert any(k.endswith("/steps10") for k in rec["workloads"]) and any(k.endswith("/steps16") for k in rec["workloads"])
