#!/bin/bash
# time one conv shape on the production library and on its compile-time ablation variants (build.py --abl MASK);
# each variant is interleaved with the baseline (same process order, same device)
# usage: tools/ablate_conv.sh "<bench_conv args>" "<masks>"
args="$1"; masks="${2:-8 16 32 64 96}"
L=self-guided-diffusion-models_amd/sgdm_amd/lib
python tools/bench_conv.py $args --reps 50
for m in $masks; do echo -n "abl=$m  "; SGDM_LIB_PATH=$L/libsgdm_hip_abl$m.so python tools/bench_conv.py $args --reps 50; done
python tools/bench_conv.py $args --reps 50
