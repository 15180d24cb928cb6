#!/usr/bin/env bash
# Build liblako_hip.so for gfx950 in-tree (the .so is git-ignored but travels to the GPU box).
#   build.sh            incremental: recompiles the sources newer than their objects
#   build.sh --force    recompiles everything (what __graft_entry__.build() runs)
#   LAKO_EXPERIMENTS=1 build.sh   → liblako_hip_exp.so with the timing experiments compiled in (tools/ only; objects *.exp.o)
#   LAKO_ASAN=1 build.sh          → liblako_hip_asan.so: the HOST code (argument validation, launch plumbing) under AddressSanitizer
#                                   (-fsanitize=address -fno-gpu-sanitize: device code is not instrumented — GPU ASan is not
#                                   available on this pool); tests/test_abi.py runs its CPU-side checks against it (objects *.asan.o)
set -euo pipefail
cd "$(dirname "$0")"
FORCE=0
[ "${1:-}" = "--force" ] && FORCE=1
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed"
OUT=../liblako_hip.so
SFX=o
LINKFLAGS=""
if [ "${LAKO_EXPERIMENTS:-0}" = "1" ]; then
  FLAGS="$FLAGS -DLAKO_EXPERIMENTS"
  OUT=../liblako_hip_exp.so
  SFX=exp.o
fi
if [ "${LAKO_ASAN:-0}" = "1" ]; then
  FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-gpu-sanitize -shared-libsan -Wno-unused-result -Wno-pass-failed"
  LINKFLAGS="-fsanitize=address -shared-libsan"
  OUT=../liblako_hip_asan.so
  SFX=asan.o
fi
objs=()
pids=()
for f in gemm rowops attn attn_enc xattn index pq bertops bertbwd comm; do
  [ -f $f.hip ] || continue
  o=$f.$SFX
  if [ $FORCE = 1 ] || [ ! -f $o ] || [ $f.hip -nt $o ] || [ common.h -nt $o ] || [ attn_shared.h -nt $o ] || [ lds_image.h -nt $o ] || [ gemm_nt4.h -nt $o ] || [ gemm_tn4.h -nt $o ] || [ gemm_nt4_mx.h -nt $o ] || [ det.h -nt $o ] || [ ../../include/lako_hip.h -nt $o ] || [ build.sh -nt $o ]; then
    extra=""
    # attention is VALU-bound on the score tiles: keep MFMA results in VGPRs (no v_accvgpr_read/write round trips)
    { [ $f = attn ] || [ $f = attn_enc ]; } && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
    $HIPCC $FLAGS $extra -c $f.hip -o $o &
    pids+=($!)
  fi
  objs+=($o)
done
for p in "${pids[@]:-}"; do if [ -n "$p" ]; then wait $p || { echo "COMPILE FAILED"; exit 1; }; fi; done
# link WITHOUT an rpath to /opt/rocm: the library must bind to the HIP runtime torch already loaded
$HIPCC --offload-arch=gfx950 -shared -fPIC $LINKFLAGS -o $OUT "${objs[@]}" -ldl
echo "built $(realpath $OUT) (${#pids[@]} source(s) compiled)"
