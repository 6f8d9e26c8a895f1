#!/usr/bin/env python3
"""Where does a training step's wall time go BETWEEN kernels?  Reads a rocprofv3 --kernel-trace CSV (one row per kernel
with start / end timestamps), cuts it into training steps at the fused optimizer kernel (adamw_ema_kernel: one per step)
and reports, for the steady-state steps: span, device-busy time (union of the kernel intervals), idle time, the number of
kernels, the idle time attributed to the kernel that FOLLOWS each gap (grouped by kernel name), and the largest gaps.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python3 $REPO/bench.py --no-extra \\
        --no-cpu-baseline --no-full --no-profile --no-train40 --no-exchange-probe --steps 2 --warmup 1
    python tools/train_timeline.py /tmp/tl/**/t_kernel_trace.csv [--marker adamw_ema_kernel]
"""
import argparse
import csv
import re
import sys
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--marker", default="adamw_ema_kernel")
ap.add_argument("--top", type=int, default=25)
a = ap.parse_args()

rows = []
with open(a.csv) as f:
    rd = csv.DictReader(f)
    for r in rd:
        name = r.get("Kernel_Name") or r.get("Name")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:70]


marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
if len(marks) < 3:
    sys.exit(f"only {len(marks)} '{a.marker}' kernels in the trace")
steps = [(marks[i] + 1, marks[i + 1] + 1) for i in range(len(marks) - 1)]
# steady state: the steps whose span is within 15 % of the median (the checksum step and the profiled pass are longer)
spans = [rows[e - 1][1] - rows[s][0] for s, e in steps]
med = sorted(spans)[len(spans) // 2]
keep = [st for st, sp in zip(steps, spans) if abs(sp - med) < 0.15 * med]
print(f"{len(steps)} steps between optimizer kernels, {len(keep)} near the median span of {med / 1e6:.2f} ms")
tot_span = tot_busy = tot_n = 0
gap_by = defaultdict(lambda: [0, 0])
dur_by = defaultdict(lambda: [0, 0])
gaps = []
for s, e in keep:
    seg = rows[s:e]
    span = seg[-1][1] - seg[0][0]
    busy, cur_end = 0, seg[0][0]
    for st, en, nm in seg:
        if st > cur_end:
            g = st - cur_end
            gap_by[short(nm)][0] += g
            gap_by[short(nm)][1] += 1
            gaps.append((g, short(nm)))
            busy += en - st
            cur_end = en
        else:
            if en > cur_end:
                busy += en - cur_end
                cur_end = en
        dur_by[short(nm)][0] += en - st
        dur_by[short(nm)][1] += 1
    tot_span += span
    tot_busy += busy
    tot_n += len(seg)
k = len(keep)
print(f"per step: span {tot_span / k / 1e6:.3f} ms, device busy {tot_busy / k / 1e6:.3f} ms, idle {(tot_span - tot_busy) / k / 1e6:.3f} ms "
      f"({100.0 * (tot_span - tot_busy) / tot_span:.1f} %), {tot_n / k:.0f} kernels, mean gap {(tot_span - tot_busy) / max(1, tot_n) / 1e3:.2f} us")
print("\nidle time in front of (per step):")
for nm, (g, n) in sorted(gap_by.items(), key=lambda kv: -kv[1][0])[:a.top]:
    print(f"  {nm:70s} {g / k / 1e3:9.1f} us over {n / k:6.1f} gaps ({g / max(1, n) / 1e3:6.2f} us each)")
print("\nkernel time (per step):")
for nm, (d, n) in sorted(dur_by.items(), key=lambda kv: -kv[1][0])[:a.top]:
    print(f"  {nm:70s} {d / k / 1e6:9.3f} ms over {n / k:6.1f} launches ({d / max(1, n) / 1e3:8.2f} us each)")
print("\nlargest gaps:")
for g, nm in sorted(gaps, reverse=True)[:12]:
    print(f"  {g / 1e3:9.1f} us before {nm}")
