#!/bin/bash
# GPU box: the evidence set of the shipped library, per BASELINE config at FULL size (c2 8192^2 6x6 -medium, c3 8192^2 8x8
# -thorough, c4 4096^2 RGBA16F HDR 6x6 -medium): rocprofv3 --kernel-trace --stats, then six separate --pmc passes
# (instruction counts, lane activity, the VALU opcode classes, FETCH_SIZE, WRITE_SIZE: the HBM passes on their own as MI355X_MICROARCH.md
# prescribes) over the same bench.py command line.  Output: gpurun_out/<tag>/ ; tools/summarize_evidence.py turns it
# into traffic.json (read back by bench.py) and a text summary.   usage: gpu_evidence.sh <tag> [configs...]
set -u
export TMPDIR=/tmp
TAG=${1:-r03e}; shift || true
CONFIGS=${@:-c2 c3 c4}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
for c in $CONFIGS; do
  CMD="python $R/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-quality --no-extra --no-host-api"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${c}_trace -o trace -- $CMD > $O/${c}_trace.log 2>&1
  find $O/${c}_trace -name '*kernel_stats*' | head -1 | xargs -r cat | cut -c1-160 | head -4
  ONE="python $R/bench.py --config $c --steps 1 --warmup 0 --no-cpu-baseline --no-quality --no-extra --no-host-api"
  timeout 600 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/${c}_pmc1 -o pmc -- $ONE > $O/${c}_pmc1.log 2>&1
  timeout 600 rocprofv3 --output-format csv --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES -d $O/${c}_pmc2 -o pmc -- $ONE > $O/${c}_pmc2.log 2>&1
  timeout 600 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH -d $O/${c}_pmc5 -o pmc -- $ONE > $O/${c}_pmc5.log 2>&1
  timeout 600 rocprofv3 --output-format csv --pmc SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES -d $O/${c}_pmc6 -o pmc -- $ONE > $O/${c}_pmc6.log 2>&1
  timeout 600 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/${c}_pmc3 -o pmc -- $ONE > $O/${c}_pmc3.log 2>&1
  timeout 600 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/${c}_pmc4 -o pmc -- $ONE > $O/${c}_pmc4.log 2>&1
  tail -c 600 $O/${c}_pmc1.log | tail -1 | cut -c1-200
done
cd $R
python tools/summarize_evidence.py $O $CONFIGS | tee $O/evidence_summary.txt
rm -f $O/*/*/*.db $O/*/*.db
du -sh $O
