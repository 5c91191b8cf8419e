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
# HBM bytes (separate passes: /opt/skills/guides/MI355X_MICROARCH.md, HBM section), LDS bank conflicts, waves in flight
for c in FETCH_SIZE WRITE_SIZE; do
  ASTCENC_AMD_LIB=$R/$LIB timeout 300 rocprofv3 --output-format csv --pmc $c -d $O/pmc_$c -o pmc -- python $R/tools/time_decode.py 8192 $B > $O/pmc_$c.log 2>&1
done
ASTCENC_AMD_LIB=$R/$LIB timeout 300 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES \
    -d $O/pmc_c -o pmc -- python $R/tools/time_decode.py 8192 $B > $O/pmc_c.log 2>&1
ASTCENC_AMD_LIB=$R/$LIB timeout 300 rocprofv3 --output-format csv --pmc SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE \
    -d $O/pmc_d -o pmc -- python $R/tools/time_decode.py 8192 $B > $O/pmc_d.log 2>&1
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
    for c, v in sorted(d.items()): print("   %-24s per dispatch %.6g" % (c, v / n[(k, c)]))
    per = lambda c: d[c] / n[(k, c)] if c in d else None
    if per("FETCH_SIZE") is not None and per("WRITE_SIZE") is not None:
        rd, wr = per("FETCH_SIZE") * 1024.0 * 2.0, per("WRITE_SIZE") * 1024.0
        print("   HBM traffic per launch: read %.6g B (2 x FETCH_SIZE KiB: gfx950 counts 128-byte requests as 64), write %.6g B, total %.6g B" % (rd, wr, rd + wr))
    if per("SQ_LDS_BANK_CONFLICT") and per("SQ_LDS_IDX_ACTIVE"):
        print("   LDS bank conflict cycles / LDS index-active cycles = %.3f" % (per("SQ_LDS_BANK_CONFLICT") / per("SQ_LDS_IDX_ACTIVE")))
    if per("SQ_WAVE_CYCLES") and per("SQ_WAVES"):
        print("   wave quad-cycles per wave %.0f; VALU %.0f, SALU %.0f, LDS %.0f instructions per wave" % tuple(
            (per(c) or 0) / per("SQ_WAVES") for c in ("SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")))
    if per("SQ_LEVEL_WAVES") and per("SQ_BUSY_CYCLES"):
        print("   mean waves in flight per SIMD (SQ_LEVEL_WAVES / SQ_BUSY_CYCLES / 4 SIMDs... see raw values): %.3g" % (per("SQ_LEVEL_WAVES") / per("SQ_BUSY_CYCLES")))
PY
rm -f $O/pmc_*/*/*.db $O/pmc_*/*.db
