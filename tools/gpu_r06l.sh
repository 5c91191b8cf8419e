#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06l; mkdir -p $O; cd $R
python -m pytest tests/test_multi_device.py -m gpu -q 2>&1 | tail -3 | tee $O/multi_device.txt
for deal in static dynamic; do
  for n in 1 8; do
    echo "== single process, $n slot(s) on one GPU, deal=$deal"; ASTC_BENCH_SHARE_GPU=1 ASTCENC_AMD_DEAL=$deal python bench.py --gpus $n --single-process --steps 2 --warmup 1 2>/dev/null | cut -c1-330
  done
done | tee $O/single_process_share_gpu.txt
