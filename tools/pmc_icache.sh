#!/bin/bash
# instruction-cache counters of the contact step (developer tool; run on the GPU box)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
C="python3 $R/bench.py --contacts --envs 65536 --steps 50 --warmup 50 --no-cpu-baseline"
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $R/gpurun_out/p_c_ic -- $C > $R/gpurun_out/p_c_ic.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/p_c_if -- $C > $R/gpurun_out/p_c_if.log 2>&1 || exit 1
echo done
