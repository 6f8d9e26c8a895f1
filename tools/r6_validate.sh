#!/bin/bash
# round 6: GPU validation of the tree -- full GPU suite + the default bench line (+ "probes": the LayerNorm routes and the HBM sweep)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${1:-r6}; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/suite.txt 2>&1; tail -4 $OUT/suite.txt
if [ "$2" = "probes" ]; then
  timeout 600 python tools/ln_hazard.py --reps 30 > $OUT/ln_hazard_f16x3.txt 2>&1
  timeout 600 python tools/ln_hazard.py --reps 30 --prec bf16x3 > $OUT/ln_hazard_bf16x3.txt 2>&1
  timeout 600 python tools/hbm_probe_sweep.py --rounds 3 > $OUT/hbm_probe.txt 2>&1
fi
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -3 $OUT/smoke.txt
timeout 1800 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; grep '^{"metric"' $OUT/bench.log | tail -1 > $OUT/bench.json
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]; t = d["train_step"]
print("ms_per_step", d["ms_per_step"], "frac", r["frac"], "ceil", r.get("frac_of_device_ceiling"), "launches", r.get("launches_per_step_all_kernels"),
      "copy", r.get("device_copy_tbps"), "train", t["ms"], "bs40", (d.get("train_step_bs40") or {}).get("ms"),
      "c5", d.get("c5", {}).get("ms_per_step"), d.get("c5", {}).get("train_step", {}).get("ms"),
      "c4", d.get("c4", {}).get("ms_per_step"), d.get("c4", {}).get("train_step", {}).get("ms"), "c2_bs80", d.get("c2_bs80", {}).get("ms_per_step"),
      "full", (d.get("full_trajectory") or {}).get("seconds"))
print({k: v["ms"] for k, v in t["roofline"]["per_kernel"].items() if v["ms"] > 0.3})
x = t.get("exchange_world1") or {}
print({k: x.get(k) for k in ("ms", "same_process_ms", "reserve_windows", "modelled_exposed_ms_at_50GBps", "modelled_exposed_ms_at_100GBps", "first_bucket_at_frac_of_backward", "exchange_backward_ms", "error")})
print([(b["mbytes"], b["enqueued_at_ms"]) for b in (x.get("exchange_buckets") or [])])
for k in ("c5", "c4", "c2_bs80"):
    rr = d.get(k, {}).get("roofline")
    if rr: print(k, "frac", rr["frac"], "ceil", rr.get("frac_of_device_ceiling"), {a: b["frac"] for a, b in rr["igemm_by_instance"].items()}, "traffic", rr.get("traffic"))
PY
