#!/bin/bash
# contact bench (65536 envs) for each variant library named on the command line (developer tool)
for v in "$@"; do
  BS_LIB=$PWD/exp_libs/$v/libbatchsim.so timeout -k 10 120 python bench.py --contacts --envs 65536 --steps 200 --warmup 50 --no-cpu-baseline > gpurun_out/v_$v.log 2>&1 || { echo "$v failed"; tail -3 gpurun_out/v_$v.log; exit 1; }
  echo "$v $(grep -o '"value": [0-9.]*' gpurun_out/v_$v.log | head -1)"
done
