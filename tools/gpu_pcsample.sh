#!/bin/bash
# GPU box: PC-sampling profile of the compression kernel (beta feature of rocprofv3).
set -u
export TMPDIR=/tmp
LIB=${1:-astc-encoder_amd/variants/libastcenc_amd_g.so}
TAG=${2:-pcs}
METHOD=${3:-host_trap}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit time --pc-sampling-interval 500 \
   --output-format csv -d /tmp/pcs_out -o pcs -- python $R/tools/time_lib.py $R/$LIB 2048 6 60 1 > $R/gpurun_out/$TAG/run.log 2>&1
echo "rc=$?"; tail -3 $R/gpurun_out/$TAG/run.log
find /tmp/pcs_out -type f | head; du -sh /tmp/pcs_out
for f in $(find /tmp/pcs_out -name '*pc_sampling*.csv'); do head -3 $f; wc -l $f; done
python $R/tools/summarize_pcsamples.py /tmp/pcs_out > $R/gpurun_out/$TAG/summary.txt 2>&1; head -60 $R/gpurun_out/$TAG/summary.txt
