#!/bin/bash
# contact bench (65536 envs) for a list of "ENV=VAL[,ENV=VAL]" settings ("-" = none); BS_LIB from the caller (developer tool)
for k in "$@"; do
  if [ "$k" = "-" ]; then e=""; else e=$(echo $k | tr ',' ' '); fi
  env $e timeout -k 10 120 python bench.py --contacts --envs 65536 --steps 200 --warmup 50 --no-cpu-baseline > gpurun_out/knob.log 2>&1 || { echo "$k failed"; tail -3 gpurun_out/knob.log; exit 1; }
  echo "$k $(grep -o '"value": [0-9.]*' gpurun_out/knob.log | head -1)"
done
