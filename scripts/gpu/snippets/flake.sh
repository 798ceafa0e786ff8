# the GPU tier three times in a row on one box (flakiness check of the stress / pipeline / rccl tests)
for i in 1 2 3; do
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -x > $O/pytest_gpu_run$i.log 2>&1
  echo "run $i: $(grep -E 'passed|failed' $O/pytest_gpu_run$i.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/pytest_gpu_run$i.log | head -5
done | tee $O/flake.txt
