#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06i; mkdir -p $O; cd $R
export ASTCENC_AMD_CACHE_DIR=/tmp/c1 ASTCENC_AMD_LOG=stderr
ASTCENC_AMD_JIT_SELF_CHECK=0 python tools/jit_debug.py 2>&1 | grep -v "amdgpu.ids\|compiled in" | tee $O/jit_vs_generic_bytes.txt | grep -v "mismatching 0 of"
python -m pytest tests/test_jit.py -m gpu -q > $O/pytest_jit.txt 2>&1; grep -n "^E  \|refused\|self-check\|passed\|failed" $O/pytest_jit.txt | cut -c1-300 | head -30
for spec in "2048 10 60" "2048 12 98" "2048 12 10"; do
  for mode in off sync; do
    echo "== $spec jit=$mode"; ASTCENC_AMD_JIT=$mode CHECK=0 python tools/time_lib.py astc-encoder_amd/libastcenc_amd.so $spec 2 2>&1 | grep -v amdgpu.ids | tail -1
  done
done 2>&1 | tee $O/jit_vs_generic_large.txt
