#!/bin/bash
# usage: tools/pmc_kernel.sh <kernel-name-substring> "<counters>" <python script args...>
# mean counter values over the launches of the kernels whose name contains the substring (one rocprofv3 --pmc pass)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
K="$1"; shift
C="$1"; shift
S="$1"; shift; case "$S" in /*) ;; *) S="$R/$S";; esac
rm -rf /tmp/pmc; rocprofv3 --pmc $C --output-format csv -d /tmp/pmc -- python3 "$S" "$@" > /tmp/pmc.log 2>&1
f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" "$K" <<PY
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: (round(sum(v)/len(v)), len(v)) for k, v in agg.items()})
PY
