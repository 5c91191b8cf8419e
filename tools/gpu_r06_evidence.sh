#!/bin/bash
# GPU box, round 6: the evidence set of the library as built (tools/gpu_r05_evidence.sh: kernel traces + PMC passes of the three
# BASELINE configs at full size, stage counts of configs 2 and 3, LDS conflicts of config 3), the per-stage wait counters of
# config 2, the decoder's kernel trace at 8192^2, the default bench line and the GPU test log.   usage: gpu_r06_evidence.sh <tag>
set -u
export TMPDIR=/tmp
TAG=${1:-r06z}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
bash tools/gpu_r05_evidence.sh $TAG > $O/evidence_all.log 2>&1; tail -5 $O/evidence_all.log
bash tools/gpu_stage_waits.sh astc-encoder_amd/variants/libastcenc_amd_dup.so ${TAG}_waits 1024 6 60 > $O/stage_waits.log 2>&1; cp gpurun_out/${TAG}_waits/stage_waits.txt $O/stage_waits_6x6_medium.txt; rm -rf gpurun_out/${TAG}_waits/dup_* gpurun_out/${TAG}_waits/lat_*
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/decode_trace -o trace -- python $R/tools/time_decode.py 8192 6 > $O/decode_trace.log 2>&1; cd $R
find $O/decode_trace -name '*kernel_stats*' | head -1 | xargs -r cat | cut -c1-200 | head -6
python tools/kernel_stats.py > $O/kernel_stats_all_builds.txt 2>&1 || true
rm -f $O/*/*/*.db $O/*/*.db
