#!/bin/bash
# kernel-trace stats of the contact bench with extra environment (developer tool): tools/kstats_env.sh TAG   (variables are taken from the caller's environment)
R=$GRAFT_REPO_ROOT
T=${1:-ks}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/$T
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T -- python3 $R/bench.py --contacts --envs 65536 --steps 50 --warmup 50 --no-cpu-baseline > $R/gpurun_out/$T.log 2>&1 || exit 1
f=$(ls $R/gpurun_out/$T/*/*kernel_stats.csv | head -1)
cut -d, -f1-4 $f | head -8
