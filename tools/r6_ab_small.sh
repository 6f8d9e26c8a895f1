#!/bin/bash
# round 6: the launcher's tile rules for small and medium launches (column_tile / want_bn256, csrc/igemm_host.hip) against the round-5
# rules (tune = NO_SMALL | ... is not expressible per tile, so: big = the 128-column tile forced, big256 = the 256-column tile forced)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r6k
V="rule=lib/libsgdm_hip.so bn128=lib/libsgdm_hip.so:tune=1 bn256=lib/libsgdm_hip.so:tune=2"
python tools/ab_conv.py --variants $V --shapes 16,256,256,8 16,512,256,8 16,128,128,8 16,128,128,16 16,256,128,16 4,512,512,16 4,256,256,32 2,128,128,64 8,512,512,16 16,256,256,16 32,256,256,16 16,512,512,16 16,256,256,32 24,512,512,16 32,512,512,16 80,256,256,8 80,512,512,8 80,1024,512,8 40,512,512,16 40,1024,512,16 80,512,512,16 --rounds 5 --reps 30 > gpurun_out/r6k/ab_small_conv.txt 2>&1
python tools/ab_conv.py --variants $V --shapes 16,256,256,8,1 16,256,768,8,1 16,128,128,16,1 4,512,512,16,1 8,512,512,16,1 4,512,1536,16,1 16,512,512,16,1 16,512,1536,16,1 80,512,1536,16,1 --rounds 5 --reps 30 > gpurun_out/r6k/ab_small_flat.txt 2>&1
cat gpurun_out/r6k/ab_small_conv.txt gpurun_out/r6k/ab_small_flat.txt | grep "^n="
python tools/bench_c1.py 2>&1 | grep -v amdgpu > gpurun_out/r6k/bench_c1.txt; cat gpurun_out/r6k/bench_c1.txt
