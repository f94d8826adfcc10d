#!/bin/bash
# per-section counters of k_tpc_chain (or k_tp_chain with --free) — developer tool, run on the GPU box
R=$GRAFT_REPO_ROOT
export BS_LIB=$R/exp_libs/cuts/libbatchsim.so
cd /tmp && export TMPDIR=/tmp
A="$1"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/cuts_a -- python3 $R/tools/cuts_run.py $A > $R/gpurun_out/cuts_a.log 2>&1 || { tail -5 $R/gpurun_out/cuts_a.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/cuts_b -- python3 $R/tools/cuts_run.py $A > $R/gpurun_out/cuts_b.log 2>&1 || { tail -5 $R/gpurun_out/cuts_b.log; exit 1; }
cd $R && python3 tools/cuts_parse.py gpurun_out/cuts_a > gpurun_out/cuts_a.txt && python3 tools/cuts_parse.py gpurun_out/cuts_b > gpurun_out/cuts_b.txt
echo done
