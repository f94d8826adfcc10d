#!/bin/bash
# headline-size contact bench (131072 envs, no gather) with extra environment (developer tool)
env $@ timeout -k 10 200 python bench.py --contacts --envs 131072 --steps 100 --warmup 50 --no-cpu-baseline > gpurun_out/v131.log 2>&1 || { echo "failed"; tail -3 gpurun_out/v131.log; exit 1; }
echo "$@ $(grep -o '"value": [0-9.]*' gpurun_out/v131.log | head -1)"
