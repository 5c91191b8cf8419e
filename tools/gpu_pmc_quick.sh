#!/bin/bash
# GPU box: instruction-mix counters for the compression kernel on a 2048^2 6x6 medium image
set -u
export TMPDIR=/tmp
LIB=${1:-astc-encoder_amd/libastcenc_amd.so}
TAG=${2:-pmcq}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
run() { n=$1; shift; rocprofv3 --output-format csv --pmc "$@" -d $R/gpurun_out/${TAG}_$n -o pmc -- python $R/tools/time_lib.py $R/$LIB 2048 6 60 1 > $R/gpurun_out/${TAG}_$n.log 2>&1; tail -1 $R/gpurun_out/${TAG}_$n.log; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM
run c SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
cd $R
python tools/summarize_pmc.py gpurun_out/${TAG}_a gpurun_out/${TAG}_b gpurun_out/${TAG}_c | tee gpurun_out/${TAG}_summary.txt
rm -f gpurun_out/${TAG}_*/*/*.db gpurun_out/${TAG}_*/*.db
