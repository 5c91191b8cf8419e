#!/bin/bash
# quick GPU iteration: parity tests (fast subset) + stage timers
set -u
mkdir -p gpurun_out
TAG=${1:-q}
timeout 900 python -m pytest tests -x -q -m gpu -k "not full_size" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest.log
python tools/prof_stages.py 2048 6 60 2>&1 | tail -18 | tee gpurun_out/${TAG}_stages.log
