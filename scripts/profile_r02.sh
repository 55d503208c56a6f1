#!/bin/bash
# Round-2 evidence run on the GPU box (via gpurun): bench lines of all four configs, rocprofv3 kernel-trace stats, HBM PMC
# passes of the headline launch, SQ PMC passes of the particle filter.  Small summaries land in gpurun_out/prof_r02/ and
# are copied into profiles/ by hand.  usage: scripts/profile_r02.sh [what...]   (what = headline others bpfpmc probes; default all)
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_r02
mkdir -p $out
what="${*:-headline others bpfpmc mfmapmc probes}"
cd /tmp && export TMPDIR=/tmp

stats() {  # stats <dir> <dest csv>: the six heaviest kernels of a kernel-trace run
  f=$(find $1 -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$2" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(sys.argv[2], "w") as g:
    w = csv.DictWriter(g, fieldnames=rows[0].keys()); w.writeheader()
    for r in rows[:8]: w.writerow(r)
for r in rows[:4]: print("   ", r["Name"][:100], r["Calls"], r["AverageNs"], r["Percentage"])
PY
}

if [[ $what == *headline* ]]; then
  echo "== headline bench (default flags)"
  python3 $root/bench.py > $out/bench_line.json 2> $out/bench_line.err; cut -c1-600 $out/bench_line.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/trace.log 2>&1
  stats $out/trace $out/bench_reference_kernel_stats.csv
  grep -h '"metric"' $out/trace.log | cut -c1-300 > $out/bench_line_profiled.json
  export PB=65536 PT=10000 PF=full5 PR=2
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/scripts/kf_one.py > $out/pmc_write.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/scripts/kf_one.py > $out/pmc_fetch.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $out/pmc_sq -- python3 $root/scripts/kf_one.py > $out/pmc_sq.log 2>&1
  python3 - <<PY
import csv, glob, json, collections
out = "$out"
summary = {"tag": "r02 headline", "launch": "scripts/kf_one.py PB=65536 PT=10000 PF=full5 (the bench's launch: FULL5, reference layout)",
           "note": "KiB per dispatch; WRITE_SIZE exact for 16-byte stores; FETCH_SIZE as reported (4-byte LDS-DMA loads: the gfx950 half-count of wide loads is not calibrated for them)"}
for name in ("pmc_write", "pmc_fetch", "pmc_sq"):
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
  rm -rf $out/trace $out/pmc_write $out/pmc_fetch $out/pmc_sq
fi

if [[ $what == *others* ]]; then
  : > $out/other_configs.jsonl
  for c in "gsf32" "gsf32 --l96-mode as_written" "gsf32 --mode collapsed" "kalman64" "bpf4096"; do
    tag=$(echo $c | tr -d ' -' )
    echo "== $c"
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_$tag -- python3 $root/bench.py --config $c --steps 3 --warmup 1 > $out/$tag.log 2>&1
    grep -h '"metric"' $out/$tag.log >> $out/other_configs.jsonl; grep -h '"metric"' $out/$tag.log | cut -c1-400
    tail -3 $out/$tag.log | grep -v metric | cut -c1-300
    stats $out/t_$tag $out/${tag}_kernel_stats.csv
    rm -rf $out/t_$tag
  done
fi

if [[ $what == *bpfpmc* ]]; then
  echo "== particle filter SQ counters (cfg4 instance, B=1024, T=50)"
  export PB=1024 PT=50 PN=4096 PREP=1
  $root/scripts/pmc_any.sh bpf_r02 bpf_scan scripts/bpf_probe.py > $out/bpf_pmc.log 2>&1
  python3 - <<PY
import json
s = json.load(open("$root/gpurun_out/pmc_bpf_r02/summary.json"))
ps = 1024 * 50 * 4096
v = s["pmc"]["SQ_INSTS_VALU"]["mean_per_dispatch"]
s["particle_steps_per_dispatch"] = ps
s["valu_wave_inst_per_particle_step_x64"] = v * 64.0 / ps
s["note"] = "SQ_INSTS_VALU counts wave64 instructions; x 64 lanes / particle-steps = VALU instructions one particle-step costs"
json.dump(s, open("$out/pmc_bpf4096.json", "w"), indent=1)
print("VALU instructions per particle-step:", s["valu_wave_inst_per_particle_step_x64"])
for k in ("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"):
    if k in s["pmc"]: print(k, s["pmc"][k]["mean_per_dispatch"])
PY
fi

if [[ $what == *mfmapmc* ]]; then
  echo "== MFMA Kalman kernel SQ counters (configs[4] instance, B=4096, T=50, no output streams / FULL5)"
  export PB=4096 PT=50
  $root/scripts/pmc_any.sh mfma_r02 kf_scan_mfma scripts/mfma_probe.py > $out/mfma_pmc.log 2>&1
  python3 - <<PY
import json
s = json.load(open("$root/gpurun_out/pmc_mfma_r02/summary.json"))
p = s["pmc"]
# SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD; GRBM_GUI_ACTIVE is summed over the 8 XCDs
if "SQ_VALU_MFMA_BUSY_CYCLES" in p and "GRBM_GUI_ACTIVE" in p:
    s["mfma_busy_frac_of_simd_cycles"] = p["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_dispatch"] / (1024.0 * p["GRBM_GUI_ACTIVE"]["mean_per_dispatch"] / 8.0)
s["note"] = "mean over the dispatches of scripts/mfma_probe.py (FULL5, FILTERED and no-stream launches of B=4096, T=50)"
json.dump(s, open("$out/pmc_kalman64.json", "w"), indent=1)
print("MFMA busy fraction of SIMD-cycles:", s.get("mfma_busy_frac_of_simd_cycles"))
for k in ("SQ_INSTS_MFMA", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU"):
    if k in p: print(k, p[k]["mean_per_dispatch"])
PY
fi

if [[ $what == *probes* ]]; then
  for p in ugsf agsf; do
    echo "== $p probe"
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_$p -- python3 $root/scripts/${p}_probe.py > $out/$p.log 2>&1
    tail -4 $out/$p.log | cut -c1-200
    stats $out/t_$p $out/${p}_kernel_stats.csv
    rm -rf $out/t_$p
  done
fi
