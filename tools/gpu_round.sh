#!/usr/bin/env bash
# One script for every GPU call of a round (replaces the per-call tools/gpu_r03*.sh one-offs).  Usage, from the repo root on the box:
#   gpurun --timeout 3600 -- 'TAG=r04z bash tools/gpu_round.sh full'       # the round-end sequence
#   gpurun --timeout 900  -- 'TAG=r04b bash tools/gpu_round.sh tests "tests/test_kernels_gpu.py -k gemm_nt" bench'
# Steps (any number, in order; a step may be followed by ONE quoted argument where noted):
#   tests "<pytest args>"   pytest -m gpu -x -q <args>            → $OUT/pytest.log
#   smoke                   __graft_entry__.smoke()               → $OUT/smoke.log
#   bench                   default `python bench.py`             → $OUT/bench.json
#   quick                   bench.py --steps 20 --warmup 5, no CPU leg, with the per-op HIP-event breakdown → $OUT/bench_quick.json, breakdown.txt
#   ab "<LAKO_TUNING a>|<LAKO_TUNING b>|…"   the quick bench once per tuning string → $OUT/ab_<i>.json (+ one summary line each)
#   ops "<bench_ops args>"  tools/bench_ops.py <args>             → $OUT/bench_ops.txt (appended)
#   trace                   rocprofv3 --kernel-trace --stats over bench.py --steps 10 --warmup 3 → kernel_stats.csv, step_timeline.txt
#   c45                     BASELINE configs 4 and 5 (T5-large, 40 / 100 passages, batch 8)
#   generate                tools/generate_probe.py
#   pmcstep                 whole-step PMC totals (three separate --pmc passes)
#   pmcgemm                 FETCH_SIZE / WRITE_SIZE of the encoder GEMM shapes at 48 000 and 64 000 rows → gemm_traffic*.json; MFMA-busy pass
#   pmcattn                 SQ / MFMA counters of the encoder attention kernels (tools/attn_probe.py)
#   dp1                     the one data-parallel number one GPU can give: the quick bench at world size 1 through RCCL (LAKO_FORCE_DIST=1) with
#                           LAKO_DP_MODE=deferred and =overlap (ticket-queue GEMMs + per-range all-reduce on a second stream), twice each → $OUT/dp1.txt
#   full                    = tests "" smoke bench trace c45 generate
set -x
TAG=${TAG:-round}
OUT=gpurun_out/$TAG
mkdir -p $OUT
QUICK="--no-cpu-baseline --all-valid-steps 0 --steps 20 --warmup 5"
step_tests() { ( time eval "timeout ${PYTEST_TIMEOUT:-2400} python -m pytest $1 -m gpu -x -q --durations=10" ) >> $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log; }
step_smoke() { python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log; }
step_bench() { ( time python bench.py ) > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json; }
step_quick() { python bench.py $QUICK --breakdown > $OUT/bench_quick.json 2> $OUT/breakdown.txt; cut -c1-220 $OUT/bench_quick.json; cat $OUT/breakdown.txt; }
step_ab() {
  local i=0
  IFS='|' read -ra TUN <<< "$1"
  for rep in 1 2; do
    for t in "${TUN[@]}"; do
      LAKO_TUNING="$t" python bench.py $QUICK > $OUT/ab_${i}.json 2> $OUT/ab_${i}.err
      echo "AB rep $rep [$t] $(python -c "import json,sys; j=json.loads(open('$OUT/ab_${i}.json').read().splitlines()[-1]); print(j['ms_per_step'], j['value'], j['roofline']['frac'])")" | tee -a $OUT/ab.txt
      i=$((i + 1))
    done
  done
}
step_ops() { python tools/bench_ops.py $1 >> $OUT/bench_ops.txt 2>&1; tail -40 $OUT/bench_ops.txt; }
step_trace() {
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --all-valid-steps 0 > $OUT/bench_under_rocprof.json 2> $OUT/prof.log
  python tools/rocpd_stats.py /tmp/prof_$TAG/b_results.db > $OUT/bench_kernel_stats.csv
  python tools/rocpd_timeline.py /tmp/prof_$TAG/b_results.db 6 > $OUT/step_timeline.txt
  python tools/rocpd_timeline.py /tmp/prof_$TAG/b_results.db 6 --all > $OUT/step_sequence.txt
  head -30 $OUT/step_timeline.txt
}
step_c45() {
  python bench.py --model large --n-passages 40 --batch 8 --no-cpu-baseline --all-valid-steps 0 --steps 15 --warmup 4 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
  python bench.py --model large --n-passages 100 --batch 8 --no-cpu-baseline --all-valid-steps 0 --steps 8 --warmup 3 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
  python bench.py --model large --n-passages 100 --batch 8 --no-cpu-baseline --all-valid-steps 0 --steps 8 --warmup 3 --fp8 > $OUT/bench_c5_fp8.json 2> $OUT/bench_c5_fp8.err
  cut -c1-200 $OUT/bench_c4.json $OUT/bench_c5.json $OUT/bench_c5_fp8.json
}
step_generate() { python tools/generate_probe.py > $OUT/generate_probe.txt 2>&1; tail -2 $OUT/generate_probe.txt; }
step_pmcstep() {
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  for p in mfma fetch write; do
    case $p in mfma) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE";; fetch) C="FETCH_SIZE";; write) C="WRITE_SIZE";; esac
    rocprofv3 --pmc $C --output-format csv -d /tmp/pmcstep/$p -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --all-valid-steps 0 > /tmp/pmcstep_$p.json 2>/dev/null
  done
  python tools/pmc_step_totals.py /tmp/pmcstep 8 | tee $OUT/step_pmc_totals.txt
}
step_pmcgemm() {
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  LAKO_PROBE_TOKENS=48000 bash tools/run_pmc.sh /tmp/pmc_g48 tools/gemm_probe.py fetch write mfma
  LAKO_PROBE_TOKENS=64000 bash tools/run_pmc.sh /tmp/pmc_g64 tools/gemm_probe.py fetch write
  python tools/gemm_traffic.py /tmp/pmc_g48 48000 > $OUT/gemm_traffic.json
  python tools/gemm_traffic.py /tmp/pmc_g64 64000 > $OUT/gemm_traffic_padded.json
  python tools/pmc_summary.py /tmp/pmc_g48 --match gemm_ > $OUT/gemm_mfma_pmc.txt 2>&1
  head -c 600 $OUT/gemm_traffic.json; tail -12 $OUT/gemm_mfma_pmc.txt
}
step_pmcattn() {
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  bash tools/run_pmc.sh /tmp/pmc_attn tools/attn_probe.py sq1 sq2 mfma
  python tools/pmc_summary.py /tmp/pmc_attn --match enc_ > $OUT/attn_pmc.txt 2>&1
  tail -30 $OUT/attn_pmc.txt
}
step_dp1() {
  for rep in 1 2; do
    for m in deferred overlap; do
      LAKO_FORCE_DIST=1 LAKO_DP_MODE=$m MASTER_ADDR=127.0.0.1 MASTER_PORT=2967$rep RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 \
        python bench.py $QUICK > $OUT/dp1_${m}_$rep.json 2> $OUT/dp1_${m}_$rep.err
      echo "world 1 through RCCL, LAKO_DP_MODE=$m rep $rep: $(python -c "import json; j=json.loads(open('$OUT/dp1_${m}_$rep.json').read().splitlines()[-1]); print(j['ms_per_step'], 'ms/step', j['value'], 'samples/s', j['config'].get('dp'))")" | tee -a $OUT/dp1.txt
    done
  done
  python bench.py $QUICK > $OUT/dp1_none.json 2>/dev/null
  echo "no process group: $(python -c "import json; j=json.loads(open('$OUT/dp1_none.json').read().splitlines()[-1]); print(j['ms_per_step'], 'ms/step')")" | tee -a $OUT/dp1.txt
}
[ $# -eq 0 ] && set -- full
while [ $# -gt 0 ]; do
  s=$1; shift
  case $s in
    tests) step_tests "$1"; shift;;
    ab) step_ab "$1"; shift;;
    ops) step_ops "$1"; shift;;
    full) step_tests ""; step_smoke; step_bench; step_trace; step_c45; step_generate;;
    smoke|bench|quick|trace|c45|generate|pmcstep|pmcgemm|pmcattn|dp1) step_$s;;
    *) echo "unknown step $s"; exit 2;;
  esac
done
