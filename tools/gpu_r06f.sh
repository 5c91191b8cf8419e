#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06f; mkdir -p $O; cd $R
python -m pytest tests/test_jit.py tests/test_decode.py tests/test_multi_device.py -m gpu -q > $O/pytest_jit_decode_multi.txt 2>&1; tail -4 $O/pytest_jit_decode_multi.txt
( export ASTCENC_AMD_CACHE_DIR=/tmp/astc_cache ASTCENC_AMD_LOG=stderr
for spec in "4096 5 60" "4096 5 98" "4096 8 10"; do
  for mode in off sync sync; do
    echo "== $spec jit=$mode"; ASTCENC_AMD_JIT=$mode CHECK=0 python tools/time_lib.py astc-encoder_amd/libastcenc_amd.so $spec 2 2>&1 | grep -v amdgpu.ids | tail -2
  done
done ) 2>&1 | tee $O/jit_vs_generic_more.txt
for deal in static dynamic static dynamic; do
  echo "== deal $deal 8192"; ASTCENC_AMD_LOG=stderr ASTCENC_AMD_DEAL=$deal ASTCENC_AMD_DEVICES=0,0,0,0,0,0,0,0 python tools/time_deal.py 8192 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $O/deal_static_vs_dynamic_8192.txt
tools/gpu_pmc_decode.sh astc-encoder_amd/libastcenc_amd.so r06f/decode_pmc 6 > $O/decode_pmc_6x6.txt 2>&1; tail -40 $O/decode_pmc_6x6.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d.get('context'), d['cpu_baseline']['value'], d['speedup_vs_cpu_baseline']); print(json.dumps(d['extra_configs'][-1])[:1500])"
