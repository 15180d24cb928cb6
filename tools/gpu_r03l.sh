# counters of round 3: encoder attention (default: persistent dQ + per-item forward / dK/dV; and all three per-item for comparison),
# and the whole training step's totals (MFMA-busy share, fabric traffic) — separate --pmc passes, counters only
set -x
OUT=gpurun_out/r03l
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/run_pmc.sh $OUT/attn_default tools/attn_probe.py sq1 sq2 mfma
python tools/pmc_summary.py $OUT/attn_default --match enc_ > $OUT/attn_pmc_default.txt 2>&1
LAKO_ATTN_PERSIST=0 bash tools/run_pmc.sh $OUT/attn_peritem tools/attn_probe.py sq1 mfma
LAKO_ATTN_PERSIST=0 python tools/pmc_summary.py $OUT/attn_peritem --match enc_ > $OUT/attn_pmc_peritem.txt 2>&1
for p in mfma fetch write; do
  case $p in mfma) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE";; fetch) C="FETCH_SIZE";; write) C="WRITE_SIZE";; esac
  rocprofv3 --pmc $C --output-format csv -d $OUT/step/$p -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --all-valid-steps 0 > $OUT/step_$p.log 2>&1
done
python tools/pmc_step_totals.py $OUT/step 8 > $OUT/step_pmc_totals.txt 2>&1
cat $OUT/attn_pmc_default.txt | head -40; cat $OUT/step_pmc_totals.txt
rm -rf $OUT/attn_default $OUT/attn_peritem $OUT/step
