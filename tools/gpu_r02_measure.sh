#!/bin/bash
# GPU box: round-2 evidence pass. (i) VALU issue-rate microbenchmark, (ii) lane-utilisation and
# instruction-class PMC passes of the compression kernel, (iii) PC-sampling attempts.
set -u
export TMPDIR=/tmp
LIB=${1:-astc-encoder_amd/libastcenc_amd.so}
TAG=${2:-r02a}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout 300 $R/tools/_build/valu_microbench 2000 > $O/valu_microbench.txt 2>&1; cat $O/valu_microbench.txt
cd /tmp
run() { n=$1; shift; timeout 300 rocprofv3 --output-format csv --pmc "$@" -d $O/pmc_$n -o pmc -- python $R/tools/time_lib.py $R/$LIB 2048 6 60 1 > $O/pmc_$n.log 2>&1; tail -1 $O/pmc_$n.log; }
run lanes SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU
run types SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_VALU
run types2 SQ_INSTS_VALU_IOPS SQ_INSTS_VALU_FLOPS_FP32 SQ_ACTIVE_INST_VALU2 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA
cd $R
python tools/summarize_pmc.py $O/pmc_lanes $O/pmc_types $O/pmc_types2 | tee $O/pmc_summary.txt
rm -f $O/pmc_*/*/*.db $O/pmc_*/*.db
# PC sampling: stochastic (hardware) first, host_trap second
cd /tmp
PLIB=${3:-astc-encoder_amd/variants/libastcenc_amd_g.so}
for METHOD in stochastic host_trap; do
  UNIT=time; INT=500
  if [ $METHOD = stochastic ]; then UNIT=cycles; INT=1048576; fi
  rm -rf /tmp/pcs_out
  timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT --pc-sampling-interval $INT \
     --output-format csv -d /tmp/pcs_out -o pcs -- python $R/tools/time_lib.py $R/$PLIB 2048 6 60 1 > $O/pcs_$METHOD.log 2>&1
  echo "pcs $METHOD rc=$?"; tail -3 $O/pcs_$METHOD.log
  for f in $(find /tmp/pcs_out -name '*pc_sampling*.csv'); do head -3 $f; wc -l $f; done
  python $R/tools/summarize_pcsamples.py /tmp/pcs_out > $O/pcs_${METHOD}_summary.txt 2>&1; head -40 $O/pcs_${METHOD}_summary.txt
done
du -sh $O
