#!/bin/bash
# Profile the headline bench on the GPU box: kernel-trace stats, then HBM PMC counters in
# separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).
# usage (via gpurun): scripts/profile_bench.sh <tag> [bench args...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $out/trace.log 2>&1
# PMC passes: the same launch (B=65536, T=10000, FULL5) through scripts/kf_one.py
export PB=65536 PT=10000 PF=full5 PR=2
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/scripts/kf_one.py > $out/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/scripts/kf_one.py > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $out/pmc_sq -- python3 $root/scripts/kf_one.py > $out/pmc_sq.log 2>&1
python3 - <<PY
import csv, glob, json, collections
out = "$out"
summary = {"tag": "$tag", "command": "bench.py --steps 5 --warmup 1 --no-cpu-baseline " + "$*"}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    summary["kernel_stats"] = [r for r in rows if "kf_scan" in r.get("Name", "")] or rows[:5]
for name in ("pmc_write", "pmc_fetch", "pmc_sq"):
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "kf_scan" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            summary.setdefault("pmc", {})[k] = {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
for f in $(find $out/trace -name "*kernel_stats.csv"); do cp $f $out/kernel_stats.csv; done
grep -h '"metric"' $out/trace.log | cut -c1-400
# keep only the small summaries in gpurun_out (the raw traces are large)
rm -rf $out/trace $out/pmc_write $out/pmc_fetch $out/pmc_sq
