#!/usr/bin/env bash
# A/B of two BUILDS of the library on the encoder's NT shapes (one gpurun call, alternating processes):
#   bash tools/gemm_lib_ab.sh lako_amd/liblako_hip.so lako_amd/liblako_hip_spread.so
for rep in 1 2; do
  for lib in "$@"; do
    echo "--- $lib (rep $rep)"
    LAKO_LIB=$PWD/$lib python tools/gemm_ab_probe.py gemm_nt_stagger 1 2>&1 | grep -v amdgpu
  done
done
