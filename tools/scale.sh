#!/usr/bin/env bash
# Weak-scaling curve of the train step on ONE node: bench.py at 1, 2, 4, 8 ranks (one process per GPU, RCCL over xGMI), the way the
# driver launches it — torch.distributed.run starts the rank processes BEFORE anything in them touches a GPU (no exec from a process
# that has initialised HIP).  One JSON line per GPU count under $OUT, and a summary: samples/s, ms/step, RCCL-reported world size,
# scaling efficiency against the 1-GPU line.
#   tools/scale.sh                                 # GPUS="1 2 4 8" STEPS=20 WARMUP=5
#   LAKO_DP_MODE=overlap LAKO_DP_GRAD_DTYPE=bf16 GPUS="2 8" tools/scale.sh
set -uo pipefail
cd "$(dirname "$0")/.."
GPUS=${GPUS:-"1 2 4 8"}
STEPS=${STEPS:-20}
WARMUP=${WARMUP:-5}
OUT=${OUT:-gpurun_out/scale}
PORT=${PORT:-29541}
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}      # dmabuf IPC: RCCL / device-tensor sharing across processes
mkdir -p "$OUT"
avail=$(python -c "import torch; print(torch.cuda.device_count())")      # (counting devices does not initialise the GPU)
for n in $GPUS; do
  if [ "$n" -gt "$avail" ]; then echo "skip N=$n: only $avail GPU(s) visible"; continue; fi
  args="--gpus $n --steps $STEPS --warmup $WARMUP --all-valid-steps 0"
  [ "$n" != 1 ] && args="$args --no-cpu-baseline"
  # `bench.py --gpus N` starts its own N ranks (torch.distributed.run as a child, before the parent touches a GPU) when no launcher
  # is around it, exactly as the driver calls it; LAKO_BENCH_PORT keeps consecutive runs on different rendezvous ports
  LAKO_BENCH_PORT=$PORT python bench.py $args --no-cpu-baseline > "$OUT/scale_n$n.json" 2> "$OUT/scale_n$n.err"
  echo "N=$n rc=$?"
  PORT=$((PORT + 1))
done
python - "$OUT" <<'PY'
import glob, json, os, sys
rows = {}
for p in glob.glob(os.path.join(sys.argv[1], "scale_n*.json")):
    lines = [ln for ln in open(p).read().splitlines() if ln.lstrip().startswith("{")]
    if lines:
        j = json.loads(lines[-1])
        rows[j["n_gpus"]] = j
base = rows.get(1)
bad = 0
for n in sorted(rows):
    j = rows[n]
    eff = f"{j['value'] / (n * base['value']):.3f}" if base else "n/a"
    ws = j['config'].get('rccl_world_size')
    print(f"N={n}: {j['value']:.1f} samples/s  {j['ms_per_step']:.2f} ms/step  rccl_world_size={ws}  "
          f"dp_mode={j['config'].get('dp_mode')} grad_dtype={j['config'].get('dp_grad_dtype')}  efficiency vs 1 GPU {eff}")
    if n > 1 and ws != n:
        print(f"  ERROR: the line for N={n} was produced by {ws} RCCL rank(s)")
        bad = 1
sys.exit(bad)
PY
