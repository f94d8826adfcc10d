#!/bin/bash
# kernel-trace stats of the contact bench (developer tool)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ks
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks -- python3 $R/bench.py --contacts --envs 65536 --steps 50 --warmup 50 --no-cpu-baseline > $R/gpurun_out/ks.log 2>&1 || exit 1
f=$(ls $R/gpurun_out/ks/*/*kernel_stats.csv | head -1)
cut -d, -f1-4 $f | head -8
