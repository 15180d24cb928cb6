#!/usr/bin/env bash
# Timing ablations of the one-pass attention backward (experiments build: wrong results by construction).  On the GPU box:
#   bash tools/attn_fused_ablate.sh > gpurun_out/<tag>/attn_fused_ablation.txt
export LAKO_LIB=$PWD/lako_amd/liblako_hip_exp.so
for spec in "0:all" "1024:no dropout hash" "32768:no bias-gradient sums" "65536:no dK/dV products" "2048:no P2" "16384:no delta stage" "8192:no ring DMA" \
            "4096:no P1 arithmetic" "6144:no P1, no P2" "14336:no P1, no P2, no DMA" "30720:barriers only"; do
  LAKO_ATTN_DEBUG=${spec%%:*} TAG="${spec#*:}" python tools/attn_time.py
done
LAKO_ATTN_FUSED_NST=3 TAG="ring of 3" python tools/attn_time.py
LAKO_ATTN_PERSIST=2 TAG="two-pass kernels (persistent dQ)" python tools/attn_time.py
