#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06k; mkdir -p $O; cd $R
python -m pytest tests/test_metrics.py -m gpu -q 2>&1 | tail -3 | tee $O/compare.txt
for lib in astc-encoder_amd/variants/libastcenc_amd_prev.so astc-encoder_amd/libastcenc_amd.so; do echo "== $lib"; ASTCENC_AMD_LIB=$R/$lib python tools/time_decode.py 8192 6 2>&1 | grep -v amdgpu.ids; done | tee -a $O/compare.txt
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python $R/tools/time_decode.py 8192 6 > $O/trace.log 2>&1; cd $R
find $O/trace -name '*kernel_stats*' | head -1 | xargs -r cat | cut -c1-200 | grep "compare\|decompress" | tee -a $O/compare.txt
rm -f $O/*/*/*.db $O/*/*.db
