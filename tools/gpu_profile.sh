#!/bin/bash
# GPU box: stage timers + rocprofv3 kernel trace + PMC passes for the bench kernel. Output in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r01}
timeout 900 python bench.py --steps 5 --warmup 1 > gpurun_out/${TAG}_bench.log 2>&1; tail -1 gpurun_out/${TAG}_bench.log
python tools/prof_stages.py 2048 6 60 > gpurun_out/${TAG}_stages_6x6_medium.log 2>&1
tail -18 gpurun_out/${TAG}_stages_6x6_medium.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-quality --no-extra --no-host-api > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace -name '*kernel_stats*' | head -1 | xargs -r cat | cut -c1-200 | head -6
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc1 -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-quality --no-extra --no-host-api > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc2 -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-quality --no-extra --no-host-api > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc2.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc3 -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-quality --no-extra --no-host-api > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc3.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc4 -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-quality --no-extra --no-host-api > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc4.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_pmc.py --json gpurun_out/${TAG}_traffic.json gpurun_out/${TAG}_pmc1 gpurun_out/${TAG}_pmc2 gpurun_out/${TAG}_pmc3 gpurun_out/${TAG}_pmc4 > gpurun_out/${TAG}_pmc_summary.txt 2>&1
cat gpurun_out/${TAG}_pmc_summary.txt
find gpurun_out/${TAG}_pmc1 gpurun_out/${TAG}_trace -type f | head -4; rm -f gpurun_out/${TAG}_*/*/*.db gpurun_out/${TAG}_*/*.db; du -sh gpurun_out
