#!/bin/bash
# usage: tools/pmc.sh "<counters>" <python script args...>   -> prints mean counter values of igemm launches
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
C="$1"; shift
S="$1"; shift; case "$S" in /*) ;; *) S="$R/$S";; esac
rm -rf /tmp/pmc; rocprofv3 --pmc $C --output-format csv -d /tmp/pmc -- python3 "$S" "$@" > /tmp/pmc.log 2>&1
f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" <<PY
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if "igemm" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v)/len(v)) for k, v in agg.items()})
PY
