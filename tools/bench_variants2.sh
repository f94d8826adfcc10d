#!/bin/bash
# contact bench (65536 envs) + contact-free bench (4096 envs) for each variant library named on the command line; "tree" = the in-tree library (developer tool)
for v in "$@"; do
  if [ "$v" = tree ]; then unset BS_LIB; else export BS_LIB=$PWD/exp_libs/$v/libbatchsim.so; fi
  timeout -k 10 120 python bench.py --contacts --envs 65536 --steps 200 --warmup 50 --no-cpu-baseline > gpurun_out/v_$v.c.log 2>&1 || { echo "$v failed"; tail -3 gpurun_out/v_$v.c.log; exit 1; }
  timeout -k 10 120 python bench.py --envs 4096 --steps 1000 --warmup 100 --no-cpu-baseline > gpurun_out/v_$v.f.log 2>&1 || { echo "$v failed"; tail -3 gpurun_out/v_$v.f.log; exit 1; }
  echo "$v contacts $(grep -o '"value": [0-9.]*' gpurun_out/v_$v.c.log | head -1) free $(grep -o '"value": [0-9.]*' gpurun_out/v_$v.f.log | head -1)"
done
