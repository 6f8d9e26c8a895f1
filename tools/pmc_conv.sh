#!/bin/bash
# usage: tools/pmc_conv.sh <lib.so or ""> "<counters>" <bench_conv args...>   -> mean counter values + duration of the igemm launches
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
LIB="$1"; shift; C="$1"; shift
[ -n "$LIB" ] && export SGDM_LIB_PATH="$R/$LIB"
rm -rf /tmp/pmc; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc -o p -- python3 $R/tools/bench_conv.py "$@" > /tmp/pmc.log 2>&1
python3 - <<PY
import csv, collections, glob
f = glob.glob("/tmp/pmc/**/p_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "igemm" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
kt = glob.glob("/tmp/pmc/**/p_kernel_trace.csv", recursive=True)[0]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt)) if "igemm" in r["Kernel_Name"]]
ns = sum(d) / len(d)
out = {k: round(sum(v) / len(v)) for k, v in agg.items()}
out["dur_us"] = round(ns / 1e3, 1)
if "GRBM_GUI_ACTIVE" in out: out["clock_GHz"] = round(out["GRBM_GUI_ACTIVE"] / 8 / ns, 3)
if "SQ_VALU_MFMA_BUSY_CYCLES" in out and "GRBM_GUI_ACTIVE" in out: out["mfma_util"] = round(out["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (out["GRBM_GUI_ACTIVE"] / 8), 3)
print(out)
PY
