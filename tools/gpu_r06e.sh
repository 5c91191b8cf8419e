#!/bin/bash
# GPU box, round 6: run-time builds (tests + timing against the generic builds), the new decoder tests, multi-device dealing.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06e; mkdir -p $O; cd $R
python -m pytest tests/test_jit.py tests/test_fixed_contexts.py tests/test_library_exports.py tests/test_decode.py tests/test_multi_device.py -m gpu -q > $O/pytest_jit_decode_multi.txt 2>&1; tail -8 $O/pytest_jit_decode_multi.txt
export ASTCENC_AMD_CACHE_DIR=/tmp/astc_cache ASTCENC_AMD_LOG=stderr
for spec in "4096 6 98" "4096 6 10" "4096 4 60" "4096 8 60" "4096 5 60" "2048 10 60" "2048 12 98"; do
  for mode in off sync sync; do
    echo "== $spec jit=$mode"; ASTCENC_AMD_JIT=$mode CHECK=0 python tools/time_lib.py astc-encoder_amd/libastcenc_amd.so $spec 2 2>&1 | grep -v amdgpu.ids | tail -2
  done
done 2>&1 | tee $O/jit_vs_generic.txt
for deal in static dynamic; do
  echo "== deal $deal"; ASTCENC_AMD_DEAL=$deal ASTCENC_AMD_DEVICES=0,0,0,0,0,0,0,0 python tools/time_deal.py 4096 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $O/deal_static_vs_dynamic.txt
