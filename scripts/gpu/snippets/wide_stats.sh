# rocprofv3 --stats of the any-width step at --hidden-size 256
cd /tmp && (timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/st_w -o s -- python $GRAFT_REPO_ROOT/bench.py --hidden-size 256 --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/stats_wide_run.log; cd $GRAFT_REPO_ROOT
find /tmp/st_w -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_hidden256.csv
head -14 $O/kernel_stats_hidden256.csv | cut -c1-200
