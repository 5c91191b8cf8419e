#!/bin/bash
# GPU box: scratch / HBM traffic counters of the compression kernel for a library variant (2048^2 6x6 medium).
set -u
export TMPDIR=/tmp
LIB=${1:-astc-encoder_amd/libastcenc_amd.so}
TAG=${2:-scratch}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
run() { n=$1; shift; timeout 300 rocprofv3 --output-format csv --pmc "$@" -d $O/pmc_$n -o pmc -- python $R/tools/time_lib.py $R/$LIB 2048 6 60 1 > $O/pmc_$n.log 2>&1; }
run a SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU
run b FETCH_SIZE
run c WRITE_SIZE
cd $R
python tools/summarize_pmc.py $O/pmc_a $O/pmc_b $O/pmc_c | tee $O/summary.txt
rm -f $O/pmc_*/*/*.db $O/pmc_*/*.db
