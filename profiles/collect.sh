#!/bin/bash
# rocprofv3 evidence for bench.py's numbers (run on the MI355X box: gpurun -- 'bash profiles/collect.sh').
# One --kernel-trace --stats run and separate --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ passes of at most 8
# counters) per workload; raw outputs land in gpurun_out/p_* (scratch), profiles/summarize.py condenses them.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --envs 4096 --steps 500 --warmup 100 --no-cpu-baseline"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_tp_stats -- $B > $R/gpurun_out/p_tp_stats.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_tp_fetch -- $B > $R/gpurun_out/p_tp_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_tp_write -- $B > $R/gpurun_out/p_tp_write.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/p_tp_sq -- $B > $R/gpurun_out/p_tp_sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/p_tp_sq2 -- $B > $R/gpurun_out/p_tp_sq2.log 2>&1 || exit 1
C="python3 $R/bench.py --contacts --envs 65536 --steps 50 --warmup 50 --no-cpu-baseline"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_c_stats -- $C > $R/gpurun_out/p_c_stats.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_c_fetch -- $C > $R/gpurun_out/p_c_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_c_write -- $C > $R/gpurun_out/p_c_write.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/p_c_sq -- $C > $R/gpurun_out/p_c_sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/p_c_sq2 -- $C > $R/gpurun_out/p_c_sq2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/p_c_mem -- $C > $R/gpurun_out/p_c_mem.log 2>&1 || exit 1
echo done
