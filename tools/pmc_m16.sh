#!/bin/bash
# Effective clock and matrix-pipe duty of the 3x3 conv kernel in its two MFMA forms (run on the GPU box, repo root):
#   tools/pmc_m16.sh gpurun_out/r3_pmc_m16.txt     (needs libsgdm_hip_nom16.so: git show 4ce7de0:profiles/r5_igemm_experiments.patch | git apply, then SGDM_BUILD_TAG=_nom16 SGDM_EXTRA_FLAGS=-DSGDM_NO_MFMA16 build.py;
#   the 128-column tile is the default for these shapes, no tile override needed)
OUT="${GRAFT_REPO_ROOT:-/root/repo}/$1"
L=self-guided-diffusion-models_amd/sgdm_amd/lib
: > "$OUT"
for shape in "--cin 512 --cout 512 --hw 16" "--cin 256 --cout 256 --hw 32" "--cin 128 --cout 128 --hw 64"; do
  for lib in "" "$L/libsgdm_hip_nom16.so"; do
    echo "== ${lib:-libsgdm_hip.so (16x16x32)} $shape" >> "$OUT"
    tools/pmc_conv.sh "$lib" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" --n 80 $shape --reps 300 >> "$OUT" 2>&1
  done
done
cat "$OUT"
