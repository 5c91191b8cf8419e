#!/bin/bash
# GPU box: LDS bank conflicts per stage, the way tools/gpu_stage_counts.sh finds instruction counts (a -DASTC_DUPSTAGE build
# runs one stage twice; the counter difference against the plain run is that stage's).  usage: gpu_stage_lds.sh <lib> <tag> [size block quality]
set -u
export TMPDIR=/tmp
LIB=${1:-astc-encoder_amd/variants/libastcenc_amd_dup.so}
TAG=${2:-stagelds}
SIZE=${3:-768}; BLOCK=${4:-8}; Q=${5:-98}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
for id in ${STAGES:-0 1 2 3 4 5 6 10 11 12 18 19 20 21 22 23 30}; do
  ASTC_DUP_STAGE=$id timeout 120 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES SQ_WAVES \
      -d $O/dup_$id -o pmc -- python $R/tools/time_lib.py $R/$LIB $SIZE $BLOCK $Q 1 > $O/dup_$id.log 2>&1
done
cd $R
python - $O <<'PY' | tee $O/stage_lds.txt
import csv, glob, os, sys
sys.path.insert(0, "tools")
d = sys.argv[1]
NAMES = {}
exec(open("tools/summarize_stage_counts.py").read().split("d = sys.argv[1]")[0].split("import csv, glob, os, sys")[1])
def load(i):
    tot = {}
    for f in glob.glob(os.path.join(d, "dup_%d" % i, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "astc_compress" in row.get("Kernel_Name", ""):
                tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    return tot
base = load(0); w = base["SQ_WAVES"]
print("per block; plain run: " + "  ".join("%s %.0f" % (k[3:], v / w) for k, v in sorted(base.items()) if k != "SQ_WAVES"))
print("%-44s %9s %9s %9s %9s" % ("stage", "LDS insts", "active", "conflict", "wavecyc"))
for i in sorted(NAMES):
    t = load(i)
    if not t: continue
    g = lambda k: (t.get(k, 0) - base.get(k, 0)) / w
    print("%-44s %9.0f %9.0f %9.0f %9.0f" % (NAMES[i], g("SQ_INSTS_LDS"), g("SQ_ACTIVE_INST_LDS"), g("SQ_LDS_BANK_CONFLICT"), g("SQ_WAVE_CYCLES")))
PY
rm -f $O/dup_*/*/*.db $O/dup_*/*.db
