# the training stream inside the pipeline: rocprofv3 kernel trace of the bench command -> tools/bench_step_trace.py
rm -rf /tmp/trb
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/trb -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-parity > /dev/null 2>&1)
(python tools/bench_step_trace.py /tmp/trb 2>&1) | tee $O/bench_step_trace.txt
