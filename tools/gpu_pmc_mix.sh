#!/bin/bash
# GPU box: dynamic instruction mix and wait breakdown of the compression kernel (three PMC passes).
# usage: gpu_pmc_mix.sh <lib> <tag> [size block quality]
set -u
export TMPDIR=/tmp
LIB=${1:-astc-encoder_amd/libastcenc_amd.so}; TAG=${2:-mix}; SIZE=${3:-2048}; BLOCK=${4:-6}; Q=${5:-60}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp
run() { name=$1; shift; CHECK=0 timeout 180 rocprofv3 --output-format csv --pmc "$@" -d $O/$name -o pmc -- python $R/tools/time_lib.py $R/$LIB $SIZE $BLOCK $Q 1 > $O/$name.log 2>&1; }
run mix SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64
run act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES
cd $R
python - <<PY | tee $O/summary.txt
import csv, glob
tot = {}
for f in glob.glob("$O/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "astc_compress" in row.get("Kernel_Name", ""):
            tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
w = tot.get("SQ_WAVES", 1.0)
for k in sorted(tot): print("%-28s %14.1f per wave" % (k, tot[k] / w))
PY
rm -f $O/*/*/*.db $O/*/*.db
