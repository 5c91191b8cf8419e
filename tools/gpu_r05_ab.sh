#!/bin/bash
# GPU box, round 5: parity of the fixed-context builds, then same-call A/B of library variants on the BASELINE shapes
# (6x6 -medium and 8x8 -thorough at 4096^2), each library with and without ASTCENC_AMD_KERNEL=generic, and the
# instruction counters of both.  usage: gpu_r05_ab.sh <tag> [lib ...]   (libs relative to the repo root)
set -u
export TMPDIR=/tmp
TAG=${1:-ab}; shift || true
LIBS=${@:-astc-encoder_amd/variants/libastcenc_amd_r04.so astc-encoder_amd/libastcenc_amd.so}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 900 python -m pytest tests/test_fixed_contexts.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
fi
for rep in 1 2; do
  for lib in $LIBS; do
    for mode in fixed generic; do
      [ "$mode" = generic ] && export ASTCENC_AMD_KERNEL=generic || unset ASTCENC_AMD_KERNEL
      case $lib in *r04*) [ "$mode" = generic ] && continue;; esac
      echo "== $lib [$mode]"
      CHECK=$([ $rep = 1 ] && echo 1 || echo 0) python tools/time_lib.py $lib 4096 6 60 3 2>&1 | tail -2
      CHECK=$([ $rep = 1 ] && echo 1 || echo 0) python tools/time_lib.py $lib 4096 8 98 2 2>&1 | tail -2
    done
  done
done 2>&1 | tee $O/ab.log
unset ASTCENC_AMD_KERNEL
if [ "${SKIP_PMC:-0}" != "1" ]; then
  cd /tmp
  for lib in $LIBS; do
    for mode in fixed generic; do
      [ "$mode" = generic ] && export ASTCENC_AMD_KERNEL=generic || unset ASTCENC_AMD_KERNEL
      case $lib in *r04*) [ "$mode" = generic ] && continue;; esac
      n=$(basename $lib .so)_$mode
      for shape in "6 60" "8 98"; do
        s=$(echo $shape | tr ' ' '_')
        CHECK=0 timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -d $O/pmc_${n}_$s -o pmc -- python $R/tools/time_lib.py $R/$lib 2048 $shape 1 > $O/pmc_${n}_$s.log 2>&1
        echo "-- $n $shape"; python $R/tools/summarize_pmc.py $O/pmc_${n}_$s | grep -v "^HBM"
      done
    done
  done 2>&1 | tee $O/pmc.log
  rm -f $O/*/*/*.db $O/*/*.db
fi
