#!/bin/bash
# GPU box: instruction counters of the decompression kernel (8192^2, given footprint).  usage: gpu_pmc_decode.sh [lib] [tag] [block]
set -u
export TMPDIR=/tmp
LIB=${1:-astc-encoder_amd/libastcenc_amd.so}
TAG=${2:-decode_pmc}
B=${3:-6}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
ASTCENC_AMD_LIB=$R/$LIB timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU \
    -d $O/pmc_a -o pmc -- python $R/tools/time_decode.py 8192 $B > $O/pmc_a.log 2>&1
ASTCENC_AMD_LIB=$R/$LIB timeout 300 rocprofv3 --output-format csv --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d $O/pmc_b -o pmc -- python $R/tools/time_decode.py 8192 $B > $O/pmc_b.log 2>&1
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(o + "/pmc_*/**/*counter_collection.csv", recursive=True) + glob.glob(o + "/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    if "decompress" not in k and "compare" not in k: continue
    print(k)
    for c, v in sorted(d.items()): print("   %-24s per dispatch %.4g" % (c, v / n[(k, c)]))
PY
rm -f $O/pmc_*/*/*.db $O/pmc_*/*.db
