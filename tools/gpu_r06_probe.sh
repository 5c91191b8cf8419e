#!/bin/bash
# GPU box, round 6 probe: (1) issue-port microbenchmark rows (tools/valu_microbench4.hip), (2) occupancy sensitivity of the
# compression kernel (run-time LDS pad of the `ldspad` experiment build), (3) wait / active counters of the product.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06a; mkdir -p $O; cd $R
tools/_build/valu_microbench4 4000 > $O/valu_microbench4.txt 2>&1; cat $O/valu_microbench4.txt
L=astc-encoder_amd/variants/libastcenc_amd_ldspad.so
for rep in 1 2; do
for pad in 0 1280 2560 3840 5120 7680 10240; do
  echo "== 6x6 medium pad $pad"; ASTC_LDS_PAD=$pad CHECK=0 python tools/time_lib.py $L 4096 6 60 2 2>&1 | tail -1
done
for pad in 0 1280 5120 10240; do
  echo "== 8x8 thorough pad $pad"; ASTC_LDS_PAD=$pad CHECK=0 python tools/time_lib.py $L 4096 8 98 1 2>&1 | tail -1
done
done 2>&1 | tee $O/lds_pad_sweep.txt
tools/gpu_pmc_mix.sh astc-encoder_amd/libastcenc_amd.so r06a/mix_c2 2048 6 60 > $O/mix_c2.txt 2>&1; tail -30 $O/mix_c2.txt
