#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06h; mkdir -p $O; cd $R
export ASTCENC_AMD_CACHE_DIR=/tmp/c1 ASTCENC_AMD_LOG=stderr
ALL="batch_refit batch_pack_hdr batch_pack_decide batch_pack_first batch_pack_retry batch_pack_retry_finish batch_score refine_quantize_candidates refine_candidate_restore refine_recompute_1partition refine_recompute_partitions refine_recompute_2planes refine_pack_hdr refine_difference refine_accept stage_ideal stage_decimate stage_angular stage_modes_1plane stage_modes_2planes stage_formats stage_partition_order stage_partition_select stage_block_statistics"
args=()
for x in $ALL; do
  list=$(for y in $ALL; do [ $y != $x ] && printf "%s," $y; done)
  args+=("-DASTC_DEBUG_LIVE_LAYOUT_IN=\"${list%,}\"")
done
python tools/jit_debug3.py 10 8 10 "${args[@]}" 2>&1 | grep -v amdgpu.ids > $O/scan_single_const.txt
i=0; for x in $ALL; do i=$((i+1)); echo "only $x constant: $(grep mismatching $O/scan_single_const.txt | sed -n ${i}p | sed 's/.*\] //')"; done | tee $O/scan_single_const_summary.txt
python -m pytest tests/test_jit.py -m gpu -q -x 2>&1 | tail -5
