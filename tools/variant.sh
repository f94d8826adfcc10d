#!/bin/bash
# Developer builds of libbatchsim.so with extra flags on some objects (profiling / cut-point builds):
#   tools/variant.sh NAME "-DNP_TIMING" batchsim k_narrow         -> exp_libs/NAME/libbatchsim.so
# Objects: batchsim k_serial k_narrow k_<tpk> w_<tpk>; the others are taken from mujoco_amd/csrc/build.
# Run with BS_LIB=exp_libs/NAME/libbatchsim.so.  exp_libs/ is git-ignored scratch.
set -e
cd "$(dirname "$0")/../mujoco_amd/csrc"
NAME=$1; FLAGS=$2; shift 2
OUT=../../exp_libs/$NAME
mkdir -p $OUT
BASE="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function"
pids=()
for o in "$@"; do
  case $o in
    batchsim|k_serial|k_narrow) src=$o.hip; def="" ;;
    k_*) src=k_tp_kernels.hip; def="-ffp-contract=on -DTPK_${o#k_}" ;;
    w_*) src=k_tp_kernels.hip; def="-ffp-contract=on -DTP_LPE=64 -DTPK_${o#w_}" ;;
  esac
  /opt/rocm/bin/hipcc $BASE $def $FLAGS -c -o $OUT/$o.o $src &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
objs=""
for f in build/*.o; do
  b=$(basename $f .o)
  if [ -f $OUT/$b.o ]; then objs="$objs $OUT/$b.o"; else objs="$objs $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libbatchsim.so $objs
echo built $OUT/libbatchsim.so
