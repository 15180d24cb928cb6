#!/usr/bin/env bash
# Build liblako_hip.so for gfx950 in-tree (the .so is git-ignored but travels to the GPU box).
set -euo pipefail
cd "$(dirname "$0")"
OUT=../liblako_hip.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed"
objs=()
pids=()
for f in gemm rowops attn index bertops; do
  [ -f $f.hip ] || continue
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ common.h -nt $f.o ] || [ ../../include/lako_hip.h -nt $f.o ]; then
    extra=""
    # attention is VALU-bound on the score tiles: keep MFMA results in VGPRs (no v_accvgpr_read/write round trips)
    [ $f = attn ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
    $HIPCC $FLAGS $extra -c $f.hip -o $f.o &
    pids+=($!)
  fi
  objs+=($f.o)
done
for p in "${pids[@]:-}"; do if [ -n "$p" ]; then wait $p || { echo "COMPILE FAILED"; exit 1; }; fi; done
# link WITHOUT an rpath to /opt/rocm: the library must bind to the HIP runtime torch already loaded
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT "${objs[@]}"
echo "built $(realpath $OUT)"
