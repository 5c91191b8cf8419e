#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06j; mkdir -p $O; cd $R
for rep in 1 2 3; do
for lib in astc-encoder_amd/variants/libastcenc_amd_prev.so astc-encoder_amd/libastcenc_amd.so; do
  CHECK=$([ $rep = 1 ] && echo 1 || echo 0) python tools/time_lib.py $lib 4096 6 60 2 2>&1 | grep -v amdgpu.ids | tail -2
done
done 2>&1 | tee $O/ab_angular_fma.txt
for lib in astc-encoder_amd/variants/libastcenc_amd_prev.so astc-encoder_amd/libastcenc_amd.so; do CHECK=1 python tools/time_lib.py $lib 2048 8 98 1 2>&1 | grep -v amdgpu.ids | tail -2; done | tee -a $O/ab_angular_fma.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py tests/test_fixed_contexts.py -m gpu -q 2>&1 | tail -3 | tee -a $O/ab_angular_fma.txt
