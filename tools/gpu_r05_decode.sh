#!/bin/bash
# GPU box, round 5 (second session): the rebuilt decoder -- its tests and fuzz on the HIP library, a same-process A/B against
# the previous decoder (variants/libastcenc_amd_r05z.so = the library of commit d9f2eb3), the rocprofv3 kernel trace of the
# decode, and the default bench line.   usage: gpu_r05_decode.sh <tag> [full]
set -u
TAG=${1:-r05d}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
T0=$SECONDS
timeout 200 python -m pytest tests/test_decode.py tests/test_volume.py tests/test_metrics.py tests/test_api_contract.py tests/test_consumer.py tests/test_photo.py tests/test_multi_device.py -m gpu -x -q -k "not eight_slots and not alpha_scale and not two_and_three" > $O/pytest_decode.txt 2>&1; echo "pytest decode rc $? ($((SECONDS - T0)) s)"; tail -3 $O/pytest_decode.txt
timeout 100 python tools/gpu_fuzz_decode.py 71 > $O/fuzz_decode.log 2>&1; echo "fuzz rc $? ($((SECONDS - T0)) s)"; tail -3 $O/fuzz_decode.log
timeout 150 python tools/time_decode_ab.py 8192 astc-encoder_amd/variants/libastcenc_amd_r05z.so astc-encoder_amd/libastcenc_amd.so > $O/decode_ab.log 2>&1; echo "ab rc $? ($((SECONDS - T0)) s)"; cat $O/decode_ab.log | tail -20
(cd /tmp && AB_CASES=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o dec -- python $R/tools/time_decode_ab.py 8192 $R/astc-encoder_amd/libastcenc_amd.so > $O/prof.log 2>&1); echo "prof rc $? ($((SECONDS - T0)) s)"
find $O/prof -name "*kernel_stats*" | head -1 | xargs -r cat | grep -i "decompress\|Name" | head -4
find $O/prof -name "*.db" -delete 2>/dev/null; find $O/prof -name "*kernel_trace*" -size +4M -delete 2>/dev/null
timeout 240 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $? ($((SECONDS - T0)) s)"; head -c 1500 $O/bench_default.json; echo
if [ "${2:-}" = "full" ]; then
  timeout 330 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest gpu rc $? ($((SECONDS - T0)) s)"; tail -3 $O/pytest_gpu.txt
fi
