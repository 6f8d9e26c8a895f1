#!/bin/bash
# round 6: GPU validation of the tree -- full GPU suite, the LayerNorm-prologue routes, the HBM probe sweep, the default bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${1:-r6c}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/suite.txt 2>&1; tail -4 $OUT/suite.txt
timeout 600 python tools/ln_hazard.py --reps 30 > $OUT/ln_hazard_f16x3.txt 2>&1; cat $OUT/ln_hazard_f16x3.txt | grep -v amdgpu.ids
timeout 600 python tools/ln_hazard.py --reps 30 --prec bf16x3 > $OUT/ln_hazard_bf16x3.txt 2>&1; cat $OUT/ln_hazard_bf16x3.txt | grep -v amdgpu.ids
timeout 600 python tools/hbm_probe_sweep.py --rounds 3 > $OUT/hbm_probe.txt 2>&1; cat $OUT/hbm_probe.txt | grep -v amdgpu.ids
timeout 1500 python bench.py > $OUT/bench.log 2>&1; grep '^{"metric"' $OUT/bench.log | tail -1 > $OUT/bench.json
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]; t = d["train_step"]
print("ms_per_step", d["ms_per_step"], "frac", r["frac"], "ceil", r.get("frac_of_device_ceiling"), "launches", r.get("launches_per_step_all_kernels"),
      "copy", r.get("device_copy_tbps"), "train", t["ms"], "bs40", (d.get("train_step_bs40") or {}).get("ms"),
      "x1", (t.get("exchange_world1") or {}).get("ms"), "c5", d.get("c5", {}).get("ms_per_step"), d.get("c5", {}).get("train_step", {}).get("ms"),
      "c4", d.get("c4", {}).get("ms_per_step"), "c2_bs80", d.get("c2_bs80", {}).get("ms_per_step"), "full", (d.get("full_trajectory") or {}).get("seconds"))
x = t.get("exchange_world1") or {}
print({k: x.get(k) for k in ("ms", "exposed_exchange_ms", "first_bucket_at_frac_of_backward", "exchange_backward_ms", "error")})
print([(b["mbytes"], b["enqueued_at_ms"]) for b in (x.get("exchange_buckets") or [])])
PY
