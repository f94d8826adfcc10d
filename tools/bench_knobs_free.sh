#!/bin/bash
# contact-free bench (4096 envs by default; ENVS=n) for a list of "ENV=VAL[,ENV=VAL]" settings ("-" = none) (developer tool)
N=${ENVS:-4096}
for k in "$@"; do
  if [ "$k" = "-" ]; then e=""; else e=$(echo $k | tr ',' ' '); fi
  env $e timeout -k 10 120 python bench.py --envs $N --steps 1000 --warmup 100 --no-cpu-baseline > gpurun_out/knobf.log 2>&1 || { echo "$k failed"; tail -3 gpurun_out/knobf.log; exit 1; }
  echo "free$N $k $(grep -o '"value": [0-9.]*' gpurun_out/knobf.log | head -1)"
done
