#!/bin/bash
# GPU box: per-stage dynamic instruction counts of the compression kernel without PC sampling.
# A -DASTC_DUPSTAGE build runs one named stage twice (wave_ctx.h: DUP_STAGE); the difference of the SQ_INSTS_*
# counters against the plain run is that stage's instruction count.  usage: gpu_stage_counts.sh <lib> <tag> [size block quality]
set -u
export TMPDIR=/tmp
LIB=${1:-astc-encoder_amd/variants/libastcenc_amd_dup.so}
TAG=${2:-stagecounts}
SIZE=${3:-1024}; BLOCK=${4:-6}; Q=${5:-60}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
for id in ${STAGES:-0 1 2 3 4 5 6 7 8 9 10 11 12 13 16 18 19 20 21 22 23}; do
  ASTC_DUP_STAGE=$id timeout 120 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES \
      -d $O/dup_$id -o pmc -- python $R/tools/time_lib.py $R/$LIB $SIZE $BLOCK $Q 1 > $O/dup_$id.log 2>&1
  echo "dup $id: $(grep -h 'Mtexels\|parity' $O/dup_$id.log | tr '\n' ' ')"
done
cd $R
python tools/summarize_stage_counts.py $O | tee $O/stage_counts.txt
rm -f $O/dup_*/*/*.db $O/dup_*/*.db
