# Round-6 closing measurements (PARTS selects: bench modes stats probes; default all).  The GPU tier + smoke and the sampler
# counters are r6_call.sh's own parts (`tests`, `pmc`).
PARTS=${PARTS:-"bench modes stats probes"}
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
line() { python - "$1" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open(f).read()); r = d["roofline"]; c = d.get("cpu_baseline") or {}
    print(f.split("/")[-1], round(d["ms_per_step"], 4), "ms/step", round(d["value"]), "subgraphs/s | windows", [round(x, 3) for x in d.get("ms_per_step_windows", [])],
          "| roofline", r["kernel"][:28], "frac", round(r["frac"], 4), "chip", r.get("frac_of_chip"), "| posemb flags", (d.get("posemb_status") or {}).get("flags"),
          "| cpu", round(c.get("value") or 0), c.get("value_is"), "| parity", list((d.get("parity") or {}).keys()))
except Exception as e:
    print(f, "FAILED", e)
PY
}
if has bench; then
  (timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench_driver.err | tail -1) > $O/bench_driver_flags.json; line $O/bench_driver_flags.json
  (timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/bench_driver.err | tail -1) > $O/bench_driver_flags_run2.json; line $O/bench_driver_flags_run2.json
  (timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/bench_driver.err | tail -1) > $O/bench_driver_flags_run3.json; line $O/bench_driver_flags_run3.json
  (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench_192.err | tail -1) > $O/bench_192_steps.json; line $O/bench_192_steps.json
  (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>>$O/bench_192.err | tail -1) > $O/bench_192_steps_run2.json; line $O/bench_192_steps_run2.json
fi
if has modes; then
  (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline --collectives 2>$O/bench_coll.err | tail -1) > $O/bench_192_steps_collectives.json; line $O/bench_192_steps_collectives.json
  (timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench_launcher.err | tail -1) > $O/bench_launcher_n1.json; line $O/bench_launcher_n1.json
  (timeout 400 python bench.py --mode e2e --no-cpu-baseline 2>$O/bench_e2e.err | tail -1) > $O/bench_e2e256.json; line $O/bench_e2e256.json
  (timeout 400 python bench.py --mode e2e --batch-size 32 --no-cpu-baseline 2>>$O/bench_e2e.err | tail -1) > $O/bench_e2e32.json; line $O/bench_e2e32.json
  (timeout 900 python bench.py --mode sample-ready --steps 192 --warmup 64 2>$O/bench_sr.err | tail -1) > $O/bench_sample_ready.json; line $O/bench_sample_ready.json
  (timeout 900 python bench.py --mode sampler --steps 96 --warmup 16 --cpu-seconds 10 2>$O/bench_g2.err | tail -1) > $O/bench_g2_sampler.json; line $O/bench_g2_sampler.json
  (timeout 300 python bench.py --mode sampler --nodes 1000000 --edges 10000000 --steps 96 --warmup 16 --cpu-seconds 10 2>$O/bench_g1s.err | tail -1) > $O/bench_g1_sampler.json; line $O/bench_g1_sampler.json
  (timeout 900 python bench.py --hidden-size 256 --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_hidden256.err | tail -1) > $O/bench_hidden256.json; line $O/bench_hidden256.json
  # soak of the multi-GPU launch path on one rank: 1024 steps with the RCCL hand-offs between the step's three graph segments
  (timeout 600 python bench.py --steps 1024 --warmup 64 --no-cpu-baseline --no-parity --collectives 2>$O/bench_coll_soak.err | tail -1) > $O/bench_1024_steps_collectives.json; line $O/bench_1024_steps_collectives.json
  python -c "import json; d=json.load(open('$O/bench_1024_steps_collectives.json')); print('collectives soak: graph_capture_failures', d.get('graph_capture_failures'), 'replays', d.get('graph_replays_in_timed_region'))"
fi
if has stats; then
  cd /tmp && (timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/st_b -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/stats_run.log; cd $GRAFT_REPO_ROOT
  find /tmp/st_b -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_default.csv
  head -8 $O/kernel_stats_default.csv | cut -c1-170
fi
if has probes; then
  (timeout 300 python tools/eval_probe.py 2>&1 | tail -8) > $O/eval_probe.txt; cut -c1-220 $O/eval_probe.txt
  (timeout 300 python tools/gin_roofline.py 2>&1 | tail -1) > $O/gin_roofline_c5.json; cut -c1-300 $O/gin_roofline_c5.json
fi
