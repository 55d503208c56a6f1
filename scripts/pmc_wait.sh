#!/bin/bash
# wait-state breakdown of the scan kernel at a given config (env for scripts/kf_one.py)
tag=$1
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmcw_$tag
mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $out/p -- python3 $GRAFT_REPO_ROOT/scripts/kf_one.py > $out/p.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES --output-format csv -d $out/q -- python3 $GRAFT_REPO_ROOT/scripts/kf_one.py > $out/q.log 2>&1
python3 - <<PY
import csv,glob,collections
for p in ("p","q"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv"%p, recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "kf_scan" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items(): print("$tag",k, "%.4g"%(sum(v)/len(v)))
PY
grep kernel= $out/p.log
rm -rf $out/p $out/q
