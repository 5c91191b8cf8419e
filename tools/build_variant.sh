#!/bin/bash
# Build an experimental variant of the product library: tools/build_variant.sh <name> "<extra hipcc flags>"
# -> astc-encoder_amd/variants/libastcenc_amd_<name>.so (git-ignored, travels with gpurun)
set -e
cd "$(dirname "$0")/../astc-encoder_amd"
mkdir -p variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-math-errno -fno-slp-vectorize -fvisibility=hidden -DASTCENC_DYNAMIC_LIBRARY=1 -Icsrc -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS $2 -shared -o variants/libastcenc_amd_$1.so csrc/kernel_ldr.hip csrc/kernel_hdr.hip csrc/kernel_ldr64.hip csrc/kernel_hdr64.hip csrc/kernel_decode.hip csrc/kernel_alpha.hip csrc/kernel_metrics.hip csrc/backend_hip.hip csrc/astcenc_entry.cpp csrc/host_tables.cpp
echo built variants/libastcenc_amd_$1.so
