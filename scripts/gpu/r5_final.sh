#!/bin/bash
# Round-5 closing measurements, one call (PARTS selects: tests pmc bench modes stats probes; default all):
#   GPU tier + smoke; sampler counters (profiles/pmc_sampler.json is keyed by the source hash) + kernel stats of the sampler alone;
#   bench lines (driver's flags with the CPU legs and the parity step, sustained, E2E, sample-ready, sampler mode on both graphs,
#   the multi-GPU launch path on one rank); rocprofv3 --stats of the bench command; eigensolver phases (with the round-4 path as
#   A/B, and the 65..128 class on its 1,024-thread workgroups: GCC_POSEMB_PAIR=0); graph probe; eval probe; wide-GIN roofline.
set -u
O=gpurun_out/${R5_OUT:-r5final}
PARTS=${PARTS:-"tests pmc bench modes stats probes"}
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
line() { python - "$1" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open(f).read()); r = d["roofline"]; c = d.get("cpu_baseline") or {}
    print(f.split("/")[-1], round(d["ms_per_step"], 4), "ms/step", round(d["value"]), "subgraphs/s | roofline", r["kernel"][:36], "frac", round(r["frac"], 4),
          "moved", r.get("frac_moved"), "| warmup", d["warmup"], "| posemb flags", (d.get("posemb_status") or {}).get("flags"), "|", (d.get("step_launch") or "")[:30],
          "| cpu", round(c.get("value") or 0), {k: round(v.get("value") or 0) for k, v in (c.get("reference_shaped") or {}).items()}, [k for k in c if k.startswith("parity")])
except Exception as e:
    print(f, "FAILED", e)
PY
}
if has tests; then
  timeout 1700 python -m pytest tests -m gpu -q --tb=short > $O/pytest_gpu.log 2>&1
  echo "== gpu tier: $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION" $O/pytest_gpu.log | head -10 | cut -c1-300
  (timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2) > $O/smoke.log; cat $O/smoke.log
fi
pmc() {  # tag, counter, args
  cd /tmp && (timeout 600 rocprofv3 --output-format csv --pmc $2 --kernel-trace -d /tmp/pmc_$1 -o p -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $3 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/log_$1.txt; cd $GRAFT_REPO_ROOT
}
stats() { # tag, args
  cd /tmp && (timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/st_$1 -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $2 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/log_st_$1.txt; cd $GRAFT_REPO_ROOT
  find /tmp/st_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_sampler_alone_$1.csv
}
if has pmc; then
  rm -f $O/pmc_sampler.json
  for S in 10 16; do
    pmc f1_$S FETCH_SIZE "--launches 24 --steps-per-call $S"
    pmc w1_$S WRITE_SIZE "--launches 24 --steps-per-call $S"
    (timeout 100 python tools/pmc_sampler.py /tmp/pmc_f1_$S /tmp/pmc_w1_$S 961441/9938200/bsz256/hops256/steps$S $O/pmc_sampler.json 2>&1 | tail -3) > $O/summary_g1_$S.log
    stats g1_steps$S "--launches 30 --steps-per-call $S"
  done
  G2="--nodes 10000000 --edges 200000000 --launches 12 --steps-per-call 16"
  pmc f2 FETCH_SIZE "$G2"
  pmc w2 WRITE_SIZE "$G2"
  (timeout 100 python tools/pmc_sampler.py /tmp/pmc_f2 /tmp/pmc_w2 9964365/199372800/bsz256/hops256/steps16 $O/pmc_sampler.json 2>&1 | tail -3) > $O/summary_g2.log
  stats g2_steps16 "$G2"
  cp $O/pmc_sampler.json profiles/pmc_sampler.json; cat $O/summary_g1_10.log $O/summary_g2.log | cut -c1-200
fi
if has bench; then
  (timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench_driver.err | tail -1) > $O/bench_driver_flags.json; line $O/bench_driver_flags.json
  (timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/bench_driver.err | tail -1) > $O/bench_driver_flags_run2.json; line $O/bench_driver_flags_run2.json
  (timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/bench_driver.err | tail -1) > $O/bench_driver_flags_run3.json; line $O/bench_driver_flags_run3.json
  (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench_192.err | tail -1) > $O/bench_192_steps.json; line $O/bench_192_steps.json
  (GCC_POSEMB_CHEB=7 timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>>$O/bench_192.err | tail -1) > $O/bench_192_steps_round4_block_solver.json; line $O/bench_192_steps_round4_block_solver.json
  (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>>$O/bench_192.err | tail -1) > $O/bench_192_steps_run2.json; line $O/bench_192_steps_run2.json
  (GCC_POSEMB_PAIR=0 GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,512,128,128 timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>>$O/bench_192.err | tail -1) > $O/bench_192_steps_1024_thread_mid_class.json; line $O/bench_192_steps_1024_thread_mid_class.json
  if [ -f gcc_amd/csrc/variants/lib_pair128.so ]; then   # the two-wave build of the 65..128 class (tools/build_variant.sh pair128 -DGCC_POSEMB_PAIR_THREADS=128)
    cp gcc_amd/csrc/libgcc_amd.so /tmp/lib_default.so; cp gcc_amd/csrc/variants/lib_pair128.so gcc_amd/csrc/libgcc_amd.so
    (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>>$O/bench_192.err | tail -1) > $O/bench_192_steps_two_wave_mid_class.json; line $O/bench_192_steps_two_wave_mid_class.json
    cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so
  fi
fi
if has modes; then
  (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline --collectives 2>$O/bench_coll.err | tail -1) > $O/bench_192_steps_collectives.json; line $O/bench_192_steps_collectives.json
  (timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench_launcher.err | tail -1) > $O/bench_launcher_n1.json; line $O/bench_launcher_n1.json
  (timeout 400 python bench.py --mode e2e --no-cpu-baseline 2>$O/bench_e2e.err | tail -1) > $O/bench_e2e256.json; line $O/bench_e2e256.json
  (timeout 400 python bench.py --mode e2e --batch-size 32 --no-cpu-baseline 2>>$O/bench_e2e.err | tail -1) > $O/bench_e2e32.json; line $O/bench_e2e32.json
  (timeout 900 python bench.py --mode sample-ready --steps 192 --warmup 64 2>$O/bench_sr.err | tail -1) > $O/bench_sample_ready.json; line $O/bench_sample_ready.json
  (timeout 900 python bench.py --mode sampler --steps 96 --warmup 16 --cpu-seconds 10 2>$O/bench_g2.err | tail -1) > $O/bench_g2_sampler.json; line $O/bench_g2_sampler.json
  (timeout 300 python bench.py --mode sampler --nodes 1000000 --edges 10000000 --steps 96 --warmup 16 --no-cpu-baseline 2>$O/bench_g1s.err | tail -1) > $O/bench_g1_sampler.json; line $O/bench_g1_sampler.json
fi
if has stats; then
  cd /tmp && (timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/st_b -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/stats_run.log; cd $GRAFT_REPO_ROOT
  find /tmp/st_b -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_default.csv
  head -8 $O/kernel_stats_default.csv | cut -c1-170
fi
if has probes; then
  (timeout 300 python tools/posemb_phases.py 2>&1 | tail -12) > $O/posemb_phases.txt; grep -E "multi call|total|^mid|^cheb|^wave" $O/posemb_phases.txt | cut -c1-260
  if [ -f gcc_amd/csrc/variants/lib_pair128.so ]; then
    cp gcc_amd/csrc/libgcc_amd.so /tmp/lib_default.so; cp gcc_amd/csrc/variants/lib_pair128.so gcc_amd/csrc/libgcc_amd.so
    (GCC_POSEMB_TEAM_SHARE=4 timeout 300 python tools/posemb_phases.py 2>&1 | tail -12) > $O/posemb_phases_two_wave_mid_class.txt; grep -E "multi call|total|^mid" $O/posemb_phases_two_wave_mid_class.txt | cut -c1-260
    cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so
  fi
  (GCC_POSEMB_PAIR=0 timeout 300 python tools/posemb_phases.py 2>&1 | tail -12) > $O/posemb_phases_1024_thread_mid_class.txt; grep -E "multi call|total|^mid" $O/posemb_phases_1024_thread_mid_class.txt | cut -c1-260
  (GCC_POSEMB_CHEB=7 timeout 300 python tools/posemb_phases.py 2>&1 | tail -12) > $O/posemb_phases_round4_block_solver.txt; grep -E "multi call|total|^cheb" $O/posemb_phases_round4_block_solver.txt | cut -c1-260
  (timeout 300 python tools/graph_probe.py --steps 200 2>&1 | tail -4) > $O/graph_probe.txt; cat $O/graph_probe.txt | cut -c1-200
  (timeout 300 python tools/eval_probe.py 2>&1 | tail -8) > $O/eval_probe.txt; cut -c1-220 $O/eval_probe.txt
  (timeout 300 python tools/gin_roofline.py 2>&1 | tail -1) > $O/gin_roofline_c5.json; cut -c1-300 $O/gin_roofline_c5.json
fi
