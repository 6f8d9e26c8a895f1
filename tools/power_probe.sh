#!/bin/bash
# Samples rocm-smi power / shader clock while one conv shape runs in a loop (is the kernel power-limited?).
#   tools/power_probe.sh "--cin 256 --cout 256 --hw 32" [extra bench_conv flags]
cd "$(dirname "$0")/.."
python tools/bench_conv.py --n 80 $1 --prec f16x3 --reps 20000 ${@:2} > /tmp/power_probe_bench.log 2>&1 &
BP=$!
sleep 20
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' '; echo
  sleep 1
done
wait $BP
tail -1 /tmp/power_probe_bench.log
