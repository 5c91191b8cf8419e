#!/bin/bash
# GPU box: VMEM store count and HBM bytes of the compression kernels on reduced-size versions of BASELINE configs 3 and 4
# (8x8 -thorough LDR at 2048^2, 6x6 -medium HDR RGBA16F at 4096^2).  usage: gpu_pmc_traffic_configs.sh [tag]
set -u
export TMPDIR=/tmp
TAG=${1:-traffic_cfg}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
run() { n=$1; cmd=$2; shift 2; timeout 300 rocprofv3 --output-format csv --pmc "$@" -d $O/$n -o pmc -- $cmd > $O/$n.log 2>&1; }
C3="python $R/tools/time_lib.py $R/astc-encoder_amd/libastcenc_amd.so 2048 8 98 1"
C4="python $R/tools/time_hdr.py 4096"
for c in c3 c4; do
  cmd=$C3; [ $c = c4 ] && cmd=$C4
  run ${c}_a "$cmd" SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VALU
  run ${c}_b "$cmd" FETCH_SIZE
  run ${c}_c "$cmd" WRITE_SIZE
  echo "== $c"; (cd $R; python tools/summarize_pmc.py $O/${c}_a $O/${c}_b $O/${c}_c) | grep -v "^W2026" | tee $O/${c}_summary.txt
done
rm -f $O/*/*/*.db $O/*/*.db
