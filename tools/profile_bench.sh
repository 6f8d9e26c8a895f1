#!/bin/bash
# rocprofv3 kernel statistics of a bench.py command (run on the GPU box, from the repo root):
#   tools/profile_bench.sh gpurun_out/r3_sampling  bench.py --steps 20 --warmup 3 --no-train --no-extra --no-full
#   tools/profile_bench.sh gpurun_out/r6_train     bench.py --steps 3 --warmup 1 --no-extra --no-full --no-train40 --no-exchange-probe
# (train runs: ALWAYS --no-exchange-probe -- the probe is a child process started from this traced, GPU-initialised parent with the
# profiler's preload in its environment: the forbidden exec-after-GPU-init pattern of this pool; bench.py also skips the probe by
# itself when it finds a rocprofv3 preload)
# (--no-full: with the 1000-step trajectory -- 1000 graph replays of ~130 kernel nodes -- inside the traced process a thread of the
# profiler dies with SIGSEGV on this image, whatever kernels the step holds; the timed K steps are the same launches)
# writes <prefix>_kernel_stats.csv (the --stats summary) and <prefix>_bench.json (the bench line of that same process).
R=${GRAFT_REPO_ROOT:-/root/repo}
PFX="$R/$1"; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o p -- python3 "$R/$1" "${@:2}" > /tmp/prof_bench.log 2>&1
grep '^{"metric"' /tmp/prof_bench.log | tail -1 > "${PFX}_bench.json"
cp "$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1)" "${PFX}_kernel_stats.csv"
head -8 "${PFX}_kernel_stats.csv" | cut -c1-200
