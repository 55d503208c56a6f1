#!/bin/bash
# usage (via gpurun): scripts/pmc_any.sh <tag> <kernel-name-substring> <python script> ; SQ counters of the matching kernel.
tag=$1; pat=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $out/p1 -- python3 $GRAFT_REPO_ROOT/$1 > $out/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d $out/p2 -- python3 $GRAFT_REPO_ROOT/$1 > $out/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $out/p3 -- python3 $GRAFT_REPO_ROOT/$1 > $out/p3.log 2>&1
python3 - <<PY
import csv,glob,collections,json
res={}
for p in ("p1","p2","p3"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv"%p, recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "$pat" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items():
            res[k]={"mean_per_dispatch": sum(v)/len(v), "dispatches": len(v)}
            print("$tag",k, sum(v)/len(v), "n=%d"%len(v))
json.dump({"tag":"$tag","kernel":"$pat","script":"$1","pmc":res}, open("$out/summary.json","w"), indent=1)
PY
rm -rf $out/p1 $out/p2 $out/p3
