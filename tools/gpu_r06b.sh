#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06b; mkdir -p $O; cd $R
for rep in 1 2; do
for lib in astc-encoder_amd/libastcenc_amd.so astc-encoder_amd/variants/libastcenc_amd_w5.so astc-encoder_amd/variants/libastcenc_amd_w5ilp.so; do
  CHECK=$([ $rep = 1 ] && echo 1 || echo 0) python tools/time_lib.py $lib 4096 6 60 2 2>&1 | tail -2
done
done 2>&1 | tee $O/ab_w5.txt
tools/gpu_stage_waits.sh astc-encoder_amd/variants/libastcenc_amd_dup.so r06b/waits 1024 6 60 2>&1 | tee $O/waits.log
