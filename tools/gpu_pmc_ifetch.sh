#!/bin/bash
# GPU box: instruction-fetch / cache counters for the compression kernel (2048^2, 6x6 medium)
set -u
export TMPDIR=/tmp
LIB=${1:-astc-encoder_amd/libastcenc_amd.so}
TAG=${2:-ifetch}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
run() { n=$1; shift; rocprofv3 --output-format csv --pmc "$@" -d $R/gpurun_out/${TAG}_$n -o pmc -- python $R/tools/time_lib.py $R/$LIB 2048 6 60 1 > $R/gpurun_out/${TAG}_$n.log 2>&1; tail -1 $R/gpurun_out/${TAG}_$n.log; }
run a SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
run b SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_MISSES
run c TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum
cd $R
python tools/summarize_pmc.py gpurun_out/${TAG}_a gpurun_out/${TAG}_b gpurun_out/${TAG}_c | tee gpurun_out/${TAG}_summary.txt
rm -f gpurun_out/${TAG}_*/*/*.db gpurun_out/${TAG}_*/*.db
