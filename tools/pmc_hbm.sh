#!/bin/bash
# HBM traffic of the bench's kernels from the memory-side L2 counters, as MI355X_MICROARCH.md "HBM" prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass), unit KiB, and on gfx950
# FETCH_SIZE counts 128-byte requests at 64 bytes -> doubled.  Output: JSON with per-launch means per kernel.
# usage (on the GPU box, from the repo root): tools/pmc_hbm.sh gpurun_out/pmc_hbm_c2.json bench.py --steps 2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT="$R/$1"; shift
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -- python3 "$R/$1" "${@:2}" > /tmp/pmc_$C.log 2>&1
done
python3 - "$OUT" "$*" <<'PY'
import collections, csv, glob, json, sys
res = collections.defaultdict(dict)


def kname(s):
    s = s.replace("void ", "").replace("(anonymous namespace)::", "")
    return s.split("<")[0].split("(")[0].strip()


for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc_{ctr}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr:
            agg[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res[k][ctr] = dict(launches=len(v), mean_kib=sum(v) / len(v))
out = {}
for k, d in res.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        rd = 2.0 * d["FETCH_SIZE"]["mean_kib"] * 1024          # gfx950 correction: x2
        wr = d["WRITE_SIZE"]["mean_kib"] * 1024
        out[k] = dict(launches=d["FETCH_SIZE"]["launches"], read_bytes_per_launch=rd, write_bytes_per_launch=wr,
                      hbm_bytes_per_launch=rd + wr)
json.dump(dict(command=sys.argv[2], method="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), KiB->bytes, "
               "FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM)", kernels=out), open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 2) for k, v in out.items()}))
PY
