#!/bin/bash
# GPU box: per-stage wait / issue-stall / active wave cycles of the compression kernel (the stage-doubling runs of
# gpu_stage_counts.sh with the SQ_WAIT_* counters), and the mean latencies of the memory instruction classes of the product.
# usage: gpu_stage_waits.sh <dup lib> <tag> [size block quality]
set -u
export TMPDIR=/tmp
LIB=${1:-astc-encoder_amd/variants/libastcenc_amd_dup.so}
TAG=${2:-stagewaits}
SIZE=${3:-1024}; BLOCK=${4:-6}; Q=${5:-60}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
for id in ${STAGES:-0 1 2 3 4 5 6 7 8 9 10 11 12 13 16 18 19 20 21 22 23}; do
  ASTC_DUP_STAGE=$id CHECK=0 timeout 120 rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU \
      -d $O/dup_$id -o pmc -- python $R/tools/time_lib.py $R/$LIB $SIZE $BLOCK $Q 1 > $O/dup_$id.log 2>&1
  echo "dup $id: $(grep -h 'Mtexels' $O/dup_$id.log | tr '\n' ' ')"
done
for m in VmemLatency LdsLatency SmemLatency InstrFetchLatency; do
  CHECK=0 timeout 120 rocprofv3 --output-format csv --pmc $m -d $O/lat_$m -o pmc -- python $R/tools/time_lib.py $R/astc-encoder_amd/libastcenc_amd.so 2048 $BLOCK $Q 1 > $O/lat_$m.log 2>&1
  echo "$m: $(grep -h astc_compress $O/lat_$m/*/*counter_collection.csv 2>/dev/null | awk -F, '{print $(NF-2), $(NF-3)}' | head -3 | tr '\n' ' ')"
done
cd $R
python - <<PY | tee $O/stage_waits.txt
import csv, glob, os
d = "$O"
def load(i):
    tot = {}
    for f in glob.glob(os.path.join(d, "dup_%d" % i, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "astc_compress" in row.get("Kernel_Name", ""):
                tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    return tot
base = load(0); w = base["SQ_WAVES"]
cols = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"]
print("per block, quad-cycles; plain run: " + "  ".join("%s %.0f" % (c[3:], base[c] / w) for c in cols))
print("%-8s %9s %9s %9s %9s %8s %7s %7s  wait%%" % ("stage", "wavecyc", "wait", "stall", "active", "VALU", "LDS", "VMEM"))
for i in [1,2,3,4,5,6,7,8,9,10,11,12,13,16,18,19,20,21,22,23]:
    t = load(i)
    if not t: continue
    dv = [(t[c] - base[c]) / w for c in cols]
    print("%-8d %9.0f %9.0f %9.0f %9.0f %8.0f %7.0f %7.0f  %5.1f" % tuple([i] + dv + [100 * dv[1] / dv[0] if dv[0] else 0]))
PY
rm -f $O/*/*/*.db $O/*/*.db
