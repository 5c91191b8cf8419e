#!/bin/bash
# GPU box: vector-memory pipeline counters (texture addresser / vL1D) for the compression kernel (2048^2, 6x6 medium)
set -u
export TMPDIR=/tmp
LIB=${1:-astc-encoder_amd/libastcenc_amd.so}
TAG=${2:-vmem}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
run() { n=$1; shift; timeout 120 rocprofv3 --output-format csv --pmc "$@" -d $R/gpurun_out/${TAG}_$n -o pmc -- python $R/tools/time_lib.py $R/$LIB 2048 6 60 1 > $R/gpurun_out/${TAG}_$n.log 2>&1; tail -2 $R/gpurun_out/${TAG}_$n.log | cut -c1-200; }
run a GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max SQ_WAVES SQ_INSTS_VMEM_RD
run b TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum
run c TD_TD_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum
run d SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
cd $R
python tools/summarize_pmc.py gpurun_out/${TAG}_a gpurun_out/${TAG}_b gpurun_out/${TAG}_c gpurun_out/${TAG}_d | tee gpurun_out/${TAG}_summary.txt
rm -f gpurun_out/${TAG}_*/*/*.db gpurun_out/${TAG}_*/*.db
