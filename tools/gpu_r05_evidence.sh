#!/bin/bash
# GPU box, round 5: the whole evidence set of the library as built -- rocprofv3 kernel stats + five PMC passes per BASELINE
# config at full size (tools/gpu_evidence.sh), the per-stage / per-trial-class instruction counts of configs 2 and 3
# (tools/gpu_stage_counts.sh with the stage-doubling build), the per-stage LDS bank conflicts of config 3
# (tools/gpu_stage_lds.sh), the static facts of every kernel build (tools/kernel_stats.py), and the two PC-sampling methods
# rocprofv3 offers (neither is supported by this stack: the error text is the record).   usage: gpu_r05_evidence.sh <tag>
set -u
TAG=${1:-r05z}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
bash tools/gpu_evidence.sh $TAG c2 c3 c4 > $O/evidence.log 2>&1; tail -70 $O/evidence.log
ALL="0 1 2 3 4 5 6 7 8 9 10 11 12 13 16 18 19 20 21 22 23 24 25 26 27 28 29 30 31"
STAGES="$ALL" bash tools/gpu_stage_counts.sh astc-encoder_amd/variants/libastcenc_amd_dup.so ${TAG}_sc_c2 1024 6 60 > $O/sc_c2.log 2>&1
cp gpurun_out/${TAG}_sc_c2/stage_counts.txt $O/stage_counts_6x6_medium.txt; tail -36 $O/stage_counts_6x6_medium.txt
STAGES="$ALL" bash tools/gpu_stage_counts.sh astc-encoder_amd/variants/libastcenc_amd_dup.so ${TAG}_sc_c3 1024 8 98 > $O/sc_c3.log 2>&1
cp gpurun_out/${TAG}_sc_c3/stage_counts.txt $O/stage_counts_8x8_thorough.txt; tail -36 $O/stage_counts_8x8_thorough.txt
bash tools/gpu_stage_lds.sh astc-encoder_amd/variants/libastcenc_amd_dup.so ${TAG}_lds_c3 768 8 98 > $O/lds_c3.log 2>&1
cp gpurun_out/${TAG}_lds_c3/stage_lds.txt $O/stage_lds_bank_conflicts_8x8_thorough.txt 2>/dev/null; tail -22 $O/stage_lds_bank_conflicts_8x8_thorough.txt
for m in host_trap stochastic; do bash tools/gpu_pcsample.sh astc-encoder_amd/libastcenc_amd.so ${TAG}_pcs_$m $m > $O/pc_sampling_$m.log 2>&1; cat gpurun_out/${TAG}_pcs_$m/run.log >> $O/pc_sampling_$m.log 2>/dev/null; tail -2 $O/pc_sampling_$m.log; done
rm -rf gpurun_out/${TAG}_sc_c2/dup_*/ gpurun_out/${TAG}_sc_c3/dup_*/ gpurun_out/${TAG}_lds_c3/dup_*/
