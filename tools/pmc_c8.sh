#!/bin/bash
# Runs ON THE GPU BOX: LDS / issue counters of one fp16-resident conv op (tools/one_c8.py; OP, WB, WC, WK, WH from the environment).
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_c8
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/tools/one_c8.py"
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -o p -- $B > $OUT/a.log 2>&1 || echo FAILED a
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/b -o p -- $B > $OUT/b.log 2>&1 || echo FAILED b
python - $OUT <<'PY'
import csv, glob, sys, collections
for leg in "ab":
    fs = glob.glob(sys.argv[1] + "/%s/**/*counter_collection.csv" % leg, recursive=True)
    if not fs:
        print("no counters for", leg); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:50]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in acc.items():
        if "c8" not in k: continue
        print(k)
        for c, v in sorted(d.items()): print("    %-28s %.4g" % (c, v))
PY
