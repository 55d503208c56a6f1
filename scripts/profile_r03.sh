#!/bin/bash
# Round-3 evidence runs on the GPU box (via gpurun).  usage: scripts/profile_r03.sh what...   (what = bpfpmc bpfstats headline others kalman64pmc ...)
# Small summaries land in gpurun_out/prof_r03/ and are copied into profiles/ by hand.
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_r03
mkdir -p $out
what="$*"
cd /tmp && export TMPDIR=/tmp

stats() {  # stats <dir> <dest csv>: the heaviest kernels of a kernel-trace run
  f=$(find $1 -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$2" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(sys.argv[2], "w") as g:
    w = csv.DictWriter(g, fieldnames=rows[0].keys()); w.writeheader()
    for r in rows[:8]: w.writerow(r)
for r in rows[:4]: print("   ", r["Name"][:110], r["Calls"], r["AverageNs"], r["Percentage"])
PY
}

pmc() {  # pmc <tag> <kernel substring> <json dest> <extra json> -- <program args...>: separate passes of a few counters each
  tag=$1; kern=$2; dest=$3; extra=$4; shift 5
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
             "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_IFETCH" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $out/pmc_${tag}_$i -- "$@" > $out/pmc_${tag}_$i.log 2>&1 || echo "pass $i failed (see $out/pmc_${tag}_$i.log)"
  done
  python3 - "$out" "$tag" "$kern" "$dest" "$extra" <<'PY'
import csv, glob, json, collections, sys
out, tag, kern, dest, extra = sys.argv[1:6]
summary = {"tag": tag, "kernel": kern, "pmc": {}}
summary.update(json.loads(extra) if extra else {})
for f in glob.glob(out + "/pmc_%s_*/**/*counter_collection.csv" % tag, recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        summary["pmc"][k] = {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
json.dump(summary, open(dest, "w"), indent=1)
print(json.dumps({k: v["mean_per_dispatch"] for k, v in summary["pmc"].items()}))
PY
  rm -rf $out/pmc_${tag}_[0-9]*
}

if [[ $what == *bpfpmc* ]]; then
  echo "== particle filter PMC (scripts/bpf_probe3.py, B=1024 T=100)"
  export PB=1024 PT=100 PREP=1 PSETS="spec=1"
  pmc bpf bpf_scan_kernel $out/pmc_bpf4096.json '{"script": "scripts/bpf_probe3.py PB=1024 PT=100 PSETS=spec=1", "particle_steps_per_dispatch": 419430400}' -- python3 $root/scripts/bpf_probe3.py
fi
if [[ $what == *bpfstats* ]]; then
  echo "== bpf4096 bench under kernel trace"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_bpf -- python3 $root/bench.py --config bpf4096 --steps 2 --warmup 1 > $out/bpf4096.log 2>&1
  grep -h '"metric"' $out/bpf4096.log > $out/bpf4096_bench_line.json; cut -c1-300 $out/bpf4096_bench_line.json
  stats $out/t_bpf $out/bpf4096_kernel_stats.csv; rm -rf $out/t_bpf
fi
for w in bf32 ugsf agsf uagsf gsf_collapsed; do
  if [[ " $what " == *" $w "* ]]; then
    echo "== $w"
    python3 $root/scripts/roofline_probe.py $w > $out/probe_$w.json 2> $out/probe_$w.err; cat $out/probe_$w.json
    kern=$(python3 -c "import json,sys; print(json.load(open('$out/probe_$w.json'))['kernel'])")
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_$w -- python3 $root/scripts/roofline_probe.py $w > $out/t_$w.log 2>&1
    stats $out/t_$w $out/${w}_kernel_stats.csv; rm -rf $out/t_$w
    pmc $w $kern $out/pmc_$w.json "$(cat $out/probe_$w.json)" -- python3 $root/scripts/roofline_probe.py $w
  fi
done

if [[ $what == *headline* ]]; then
  echo "== headline bench (default flags)"
  python3 $root/bench.py > $out/bench_line.json 2> $out/bench_line.err; cut -c1-600 $out/bench_line.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/trace.log 2>&1
  stats $out/trace $out/bench_reference_kernel_stats.csv
  grep -h '"metric"' $out/trace.log | cut -c1-300 > $out/bench_line_profiled.json
  export PB=65536 PT=10000 PF=full5 PR=2
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/scripts/kf_one.py > $out/pmc_write.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/scripts/kf_one.py > $out/pmc_fetch.log 2>&1
  python3 - <<PY
import csv, glob, json, collections
out = "$out"
summary = {"tag": "r03 headline", "launch": "scripts/kf_one.py PB=65536 PT=10000 PF=full5 (the bench's launch: FULL5, reference layout)",
           "note": "KiB per dispatch; WRITE_SIZE exact for 16-byte stores; FETCH_SIZE as reported (4-byte LDS-DMA loads: the gfx950 half-count of wide loads is not calibrated for them)"}
for name in ("pmc_write", "pmc_fetch"):
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "kf_scan" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            summary.setdefault("pmc", {})[k] = {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
json.dump(summary, open(out + "/bench_reference_summary.json", "w"), indent=1)
print(json.dumps(summary.get("pmc", {}), indent=0)[:600])
PY
  rm -rf $out/trace $out/pmc_write $out/pmc_fetch
fi

if [[ $what == *others* ]]; then
  : > $out/other_configs.jsonl
  for c in "gsf32 --l96-mode as_written" "gsf32" "gsf32 --mode collapsed" "kalman64" "bpf4096"; do
    tag=$(echo $c | tr -d ' -' )
    echo "== $c"
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_$tag -- python3 $root/bench.py --config $c --steps 3 --warmup 1 > $out/$tag.log 2>&1
    grep -h '"metric"' $out/$tag.log >> $out/other_configs.jsonl; grep -h '"metric"' $out/$tag.log | cut -c1-400
    tail -3 $out/$tag.log | grep -v metric | cut -c1-300
    stats $out/t_$tag $out/${tag}_kernel_stats.csv
    rm -rf $out/t_$tag
  done
fi

if [[ $what == *gsf32pmc* ]]; then
  echo "== gsf32 (as_written) HBM traffic per 500-step chunk"
  for c in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/pmc_gsf32_$c -- python3 $root/bench.py --config gsf32 --l96-mode as_written --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_gsf32_$c.log 2>&1
  done
  python3 - <<PY
import csv, glob, json, collections
out = "$out"
summary = {"tag": "r03 gsf32 as_written", "launch": "bench.py --config gsf32 --l96-mode as_written --steps 1 --warmup 0 (10 chunks of 500 steps, B = 16384, K = 32, FULL5)",
           "algorithmic_bytes_per_dispatch": 18576 * 16384 * 500, "note": "KiB per dispatch of gsf_scan_kernel"}
for c in ("WRITE_SIZE", "FETCH_SIZE"):
    for f in glob.glob(out + "/pmc_gsf32_%s/**/*counter_collection.csv" % c, recursive=True):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gsf_scan_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
        if v: summary.setdefault("pmc", {})[c] = {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
json.dump(summary, open(out + "/pmc_gsf32_traffic.json", "w"), indent=1)
print(json.dumps(summary.get("pmc", {})))
PY
  rm -rf $out/pmc_gsf32_WRITE_SIZE $out/pmc_gsf32_FETCH_SIZE
fi
if [[ $what == *kalman64pmc* ]]; then
  echo "== kalman64 SQ counters (bench launch: B = 32768, 100-step chunks)"
  pmc kalman64 kf_scan_mfma5_kernel $out/pmc_kalman64.json '{"script": "bench.py --config kalman64 --steps 1 --warmup 0", "steps_per_dispatch": 3276800}' -- python3 $root/bench.py --config kalman64 --steps 1 --warmup 0 --no-cpu-baseline
fi
