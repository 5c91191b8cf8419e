#!/bin/bash
# GPU box: the run-time specialised builds -- their tests, and kernel time of a context on its run-time build against the
# generic build of the library (same process order, same image).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06d; mkdir -p $O; cd $R
python -m pytest tests/test_jit.py tests/test_fixed_contexts.py tests/test_library_exports.py -m gpu -x -q > $O/pytest_jit.txt 2>&1; tail -5 $O/pytest_jit.txt
export ASTCENC_AMD_CACHE_DIR=/tmp/astc_cache ASTCENC_AMD_LOG=stderr
for spec in "4096 6 98" "4096 6 10" "4096 4 60" "4096 8 60" "4096 5 60" "2048 10 60" "4096 8 98" "4096 6 60"; do
  for mode in off sync sync; do
    echo "== $spec jit=$mode"; ASTCENC_AMD_JIT=$mode CHECK=0 python tools/time_lib.py astc-encoder_amd/libastcenc_amd.so $spec 2 2>&1 | grep -v amdgpu.ids | tail -2
  done
done 2>&1 | tee $O/jit_vs_generic.txt
