#!/bin/bash
# usage: tools/pmc.sh <outdir-name> <counters...> -- <bench args>
# Collects PMC counters for the bench kernel (own run: kernel-trace only, no other trace domains).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp
rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d $R/gpurun_out/$name -- python3 $R/bench.py "$@" > $R/gpurun_out/$name.log 2>&1
f=$(find $R/gpurun_out/$name -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "reconstruct" not in k: continue
    print(k)
    for c, v in d.items():
        print(f"   {c:28s} n={len(v):4d} mean={sum(v)/len(v):14.1f}")
PY
