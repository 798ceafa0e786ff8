# generate.py's eval-mode encoder: the 15-launch chain against the one-launch kernel at rw_hops 64 and 256
( timeout 300 python tools/eval_probe.py 2>&1 | tail -5
  timeout 300 python tools/eval_probe.py --rw-hops 256 --nodes 100000 --edges 1000000 2>&1 | tail -5 ) | grep -v amdgpu.ids | tee $O/eval_probe.txt
