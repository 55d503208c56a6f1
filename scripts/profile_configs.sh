#!/bin/bash
# rocprofv3 kernel-trace stats for the extra bench configs (cfg3/4/5 shapes).  usage: scripts/profile_configs.sh <tag>
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/profc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in gsf32 kalman64 bpf4096; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$c -- python3 $root/bench.py --config $c --steps 2 --warmup 1 > $out/$c.log 2>&1
  f=$(find $out/$c -name "*kernel_stats.csv" | head -1)
  echo "== $c"; grep -h '"metric"' $out/$c.log | cut -c1-200
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
with open("$out/${c}_kernel_stats.csv","w") as g:
    w=csv.DictWriter(g, fieldnames=rows[0].keys()); w.writeheader()
    for r in rows[:6]: w.writerow(r)
for r in rows[:3]: print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
  rm -rf $out/$c
done
