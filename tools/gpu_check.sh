#!/bin/bash
# Runs on the GPU box via gpurun: smoke, GPU parity tests, a short bench, and a rocprofv3 kernel trace.
# Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/summary.log
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/summary.log
tail -5 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/summary.log
tail -2 gpurun_out/bench.log
