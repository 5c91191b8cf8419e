#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06n; mkdir -p $O; cd $R
for s in 64 256 512 1024 1536 2048 3072 4096 8192; do python tools/time_host_api.py $s 6 60 2>&1 | grep -v amdgpu.ids | tail -1; done | tee $O/host_api_small_images.txt
python tools/corpus_rates.py 6x6 medium 5 2>&1 | tee $O/corpus_rates_6x6_medium.txt | tail -16
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
