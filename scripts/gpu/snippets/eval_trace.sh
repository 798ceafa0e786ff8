# per-kernel durations of the eval paths: rocprofv3 kernel trace of tools/eval_probe.py at rw_hops 64 and 256
for cfg in "64 --nodes 10000 --edges 100000" "256 --nodes 100000 --edges 1000000"; do
  set -- $cfg
  rm -rf /tmp/tre
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tre -o t -- python $GRAFT_REPO_ROOT/tools/eval_probe.py --rw-hops $cfg --reps 20 > /dev/null 2>&1)
  echo "rw_hops $1" | tee -a $O/eval_trace.txt
  f=$(find /tmp/tre -name '*kernel_stats.csv' | head -1)
  head -14 "$f" | cut -c1-200 | tee -a $O/eval_trace.txt
done
