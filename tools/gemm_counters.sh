#!/bin/bash
# SQ counters of the largest product of the backward pass (G0 . W: 9728 x 512 x 10000 on two scaled f16 pieces) run alone,
# two passes of eight counters each, per launch.    bash tools/gemm_counters.sh [path of another build of the library]
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  i=$((i+1)); rm -rf /tmp/pg$i
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d /tmp/pg$i -o x -- python3 $root/tools/one_gemm.py $1 > /tmp/pg$i.log 2>&1 || { tail -5 /tmp/pg$i.log; exit 1; }
  python3 - /tmp/pg$i <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    kn = r["Kernel_Name"]
    if "gemm_f16x2s" not in kn: continue
    acc[kn][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == list(acc[kn].keys())[0]: cnt[kn] += 1
for kn in acc:
    print(kn[:60], "launches", cnt[kn])
    for c, v in acc[kn].items(): print("   %-28s %.4g per launch" % (c, v / cnt[kn]))
PY
done
