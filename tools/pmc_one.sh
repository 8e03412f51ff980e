#!/bin/bash
# SQ counters of the kernels of one python tool: bash tools/pmc_one.sh <out name> <kernel substring> <tool.py args...>
set -e
OUT=$1; FLT=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/$OUT.tmp
rm -rf $D && mkdir -p $D
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $D/a -- python3 "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --kernel-trace -d $D/b -- python3 "$@" > /dev/null 2>&1
python3 tools/pmc_kernels.py $D/a "$FLT" > gpurun_out/$OUT.txt 2>&1
python3 tools/pmc_kernels.py $D/b "$FLT" >> gpurun_out/$OUT.txt 2>&1
rm -rf $D
cat gpurun_out/$OUT.txt
