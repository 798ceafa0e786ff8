#!/bin/bash
# Round 6: ONE parametrised call script (replaces per-call r5_callN.sh files).  Usage:
#   gpurun -- 'bash scripts/gpu/r6_call.sh <out-tag> <part> [<part> ...]'
# parts (each writes under gpurun_out/<out-tag>/):
#   stream        tools/graph_probe.py + rocprofv3 launch trace -> tools/stream_trace.py
#   ginphases     tools/gin_phases.py
#   tests         whole GPU tier + smoke
#   tests:<expr>  pytest -m gpu -k <expr>
#   phases        tools/posemb_phases.py
#   bench|driver  sustained (192 steps) / driver flags (20 steps), --no-cpu-baseline
#   variant:<n>   swap gcc_amd/csrc/variants/lib_<n>.so in for the parts that follow (variant:default swaps back)
#   sh:<file>     source an extra snippet (scripts/gpu/snippets/<file>.sh) with $O set
set -u
tag=$1; shift
O=gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
cp gcc_amd/csrc/libgcc_amd.so /tmp/lib_default.so
v=default
for part in "$@"; do
  case $part in
    variant:*) v=${part#variant:}
      if [ $v = default ]; then cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so; else cp gcc_amd/csrc/variants/lib_$v.so gcc_amd/csrc/libgcc_amd.so; fi ;;
    stream) (timeout 300 python tools/graph_probe.py --steps 200 2>&1 | tail -3) > $O/graph_probe_$v.txt
      rm -rf /tmp/tr_$v
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$v -o t -- python $GRAFT_REPO_ROOT/tools/graph_probe.py --steps 60 > /dev/null 2>&1)
      (python tools/stream_trace.py /tmp/tr_$v 2>&1) > $O/stream_trace_$v.txt
      echo "[$v] $(tail -n 2 $O/graph_probe_$v.txt)"; sed -n '/^busy/,$p' $O/stream_trace_$v.txt | head -30 ;;
    ginphases) (timeout 300 python tools/gin_phases.py 2>&1 | tail -4) > $O/gin_phases_$v.txt; echo "[$v]"; cat $O/gin_phases_$v.txt ;;
    tests) timeout 1700 python -m pytest tests -m gpu -q --tb=short > $O/pytest_gpu_$v.log 2>&1
      echo "[$v] gpu tier: $(grep -E 'passed|failed' $O/pytest_gpu_$v.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION" $O/pytest_gpu_$v.log | head -10 | cut -c1-300
      (timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2) > $O/smoke_$v.log; cat $O/smoke_$v.log ;;
    tests:*) k=${part#tests:}
      timeout 1200 python -m pytest tests -m gpu -q --tb=short -rP -k "$k" > $O/pytest_${k// /_}_$v.log 2>&1
      echo "[$v] tests -k '$k': $(grep -E 'passed|failed' $O/pytest_${k// /_}_$v.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION|^E  " $O/pytest_${k// /_}_$v.log | head -20 | cut -c1-300 ;;
    phases) (timeout 300 python tools/posemb_phases.py 2>&1 | grep -E "multi call|^mid|^cheb|^wave|^total|status") > $O/posemb_phases_$v.txt
      echo "[$v]"; cut -c1-250 $O/posemb_phases_$v.txt ;;
    bench|driver)
      if [ $part = bench ]; then flags="--steps 192 --warmup 64"; f=bench_192; else flags="--steps 20 --warmup 5"; f=bench_driver; fi
      (timeout 400 python bench.py $flags --no-cpu-baseline --no-parity 2>>$O/bench_$v.err | tail -1) > $O/${f}_$v.json
      python -c "
import json; d=json.loads(open('$O/${f}_$v.json').read()); s=d['stage_rooflines']; print('[$v] $f', round(d['ms_per_step'],4), 'ms/step', round(d['value']), 'subgraphs/s flags', (d.get('posemb_status') or {}).get('flags'), 'encoder fwd/bwd in step', round(s['gin_encoder_fwd']['ms_in_step'],3), round(s['gin_encoder_bwd']['ms_in_step'],3))" ;;
    pmc)  # sampler counters (separate --pmc passes, as MI355X_MICROARCH.md prescribes) + kernel stats of the sampler alone -> profiles/pmc_sampler.json
      pmc_pass() { cd /tmp && (timeout 600 rocprofv3 --output-format csv --pmc $2 --kernel-trace -d /tmp/pmc_$1 -o p -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $3 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/log_$1.txt; cd $GRAFT_REPO_ROOT; }
      st_pass() { cd /tmp && (timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/st_$1 -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $2 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/log_st_$1.txt; cd $GRAFT_REPO_ROOT
                  find /tmp/st_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_sampler_alone_$1.csv; }
      rm -f $O/pmc_sampler.json
      for S in 10 16; do
        pmc_pass f1_$S FETCH_SIZE "--launches 24 --steps-per-call $S"
        pmc_pass w1_$S WRITE_SIZE "--launches 24 --steps-per-call $S"
        (timeout 100 python tools/pmc_sampler.py /tmp/pmc_f1_$S /tmp/pmc_w1_$S 961441/9938200/bsz256/hops256/steps$S $O/pmc_sampler.json 2>&1 | tail -3) > $O/summary_g1_$S.log
        st_pass g1_steps$S "--launches 30 --steps-per-call $S"
      done
      G2="--nodes 10000000 --edges 200000000 --launches 12 --steps-per-call 16"
      pmc_pass f2 FETCH_SIZE "$G2"
      pmc_pass w2 WRITE_SIZE "$G2"
      (timeout 100 python tools/pmc_sampler.py /tmp/pmc_f2 /tmp/pmc_w2 9964365/199372800/bsz256/hops256/steps16 $O/pmc_sampler.json 2>&1 | tail -3) > $O/summary_g2.log
      st_pass g2_steps16 "$G2"
      cat $O/summary_g1_10.log $O/summary_g2.log | cut -c1-200 ;;
    sh:*) . scripts/gpu/snippets/${part#sh:}.sh ;;
    *) echo "unknown part $part" ;;
  esac
done
cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so
