#!/bin/bash
# round 6: the rocprofv3 summaries and PMC passes the bench line's roofline refers to (run on the GPU box from the repo root)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${1:-r6p}; mkdir -p $OUT
bash tools/profile_bench.sh $OUT/r6_bench_c2_sampling bench.py --steps 20 --warmup 3 --no-train --no-extra --no-full --no-cpu-baseline
mv $OUT/r6_bench_c2_sampling_bench.json $OUT/r6_bench_c2_sampling_under_rocprof.json
bash tools/profile_bench.sh $OUT/r6_bench_c2_train bench.py --steps 3 --warmup 1 --no-extra --no-full --no-train40 --no-exchange-probe --no-cpu-baseline
mv $OUT/r6_bench_c2_train_bench.json $OUT/r6_bench_c2_train_under_rocprof.json
COMMON="--steps 2 --warmup 1 --no-train --no-extra --no-train40 --no-exchange-probe --no-full --no-cpu-baseline"
bash tools/pmc_hbm.sh $OUT/r6_pmc_hbm_c2.json bench.py $COMMON
bash tools/pmc_hbm.sh $OUT/r6_pmc_hbm_c2_bs80.json bench.py --batch 80 $COMMON
bash tools/pmc_hbm.sh $OUT/r6_pmc_hbm_c5.json bench.py --workload c5 $COMMON
bash tools/pmc_hbm.sh $OUT/r6_pmc_hbm_c4.json bench.py --workload c4 $COMMON
ls -la $OUT
