#!/bin/bash
# GPU box: quick loop of the decoder work -- decoder tests and fuzz on the product library, then tools/time_decode_ab.py over
# the libraries named (all cases), then over a second list with the first case only (phase-split builds).
# usage: gpu_r05_decode_ab.sh <tag> "<libs, all cases>" "<libs, first case>"
set -u
TAG=${1:-r05e}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
T0=$SECONDS
timeout 200 python -m pytest tests/test_decode.py tests/test_volume.py tests/test_metrics.py -m gpu -x -q > $O/pytest_decode.txt 2>&1; echo "pytest decode rc $? ($((SECONDS - T0)) s)"; tail -3 $O/pytest_decode.txt
timeout 100 python tools/gpu_fuzz_decode.py 72 > $O/fuzz_decode.log 2>&1; echo "fuzz rc $? ($((SECONDS - T0)) s)"; tail -3 $O/fuzz_decode.log
timeout 150 python tools/time_decode_ab.py 8192 $2 > $O/decode_ab.log 2>&1; echo "ab rc $? ($((SECONDS - T0)) s)"; grep -v amdgpu.ids $O/decode_ab.log | tail -30
if [ -n "${3:-}" ]; then
  AB_CASES=1 timeout 100 python tools/time_decode_ab.py 8192 $3 > $O/decode_phases.log 2>&1; echo "phases rc $? ($((SECONDS - T0)) s)"; grep -v amdgpu.ids $O/decode_phases.log | tail -12
fi
