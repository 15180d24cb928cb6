mkdir -p gpurun_out/r02l
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for p in mfma fetch write; do
  case $p in mfma) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE";; fetch) C="FETCH_SIZE";; write) C="WRITE_SIZE";; esac
  rocprofv3 --pmc $C --output-format csv -d /tmp/pmcstep/$p -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --all-valid-steps 0 > /tmp/pmcstep_$p.json 2>/dev/null
  cut -c1-160 /tmp/pmcstep_$p.json
done
python tools/pmc_step_totals.py /tmp/pmcstep 8 | tee gpurun_out/r02l/step_pmc_totals.txt
