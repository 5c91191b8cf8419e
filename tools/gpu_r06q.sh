#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R
for rep in 1 2; do
for lib in astc-encoder_amd/libastcenc_amd.so astc-encoder_amd/variants/libastcenc_amd_os.so astc-encoder_amd/variants/libastcenc_amd_o2.so astc-encoder_amd/variants/libastcenc_amd_nounroll.so; do
  CHECK=$([ $rep = 1 ] && echo 1 || echo 0) python tools/time_lib.py $lib 4096 6 60 2 2>&1 | grep -v amdgpu.ids | tail -2
done
done 2>&1 | tee $O/ab_code_size.txt
