#!/usr/bin/env bash
# rocprofv3 --pmc passes (counters only: no trace domains beside them) over a probe script, one pass per counter group.
#   tools/run_pmc.sh <out-dir> <probe.py> [pass names…]     passes: sq1 sq2 mfma fetch write
set -uo pipefail
OUT=$1; PROBE=$2; shift 2
PASSES=${*:-"sq1 sq2 mfma"}
export TMPDIR=/tmp
declare -A C
C[sq1]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
C[sq2]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
C[mfma]="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
C[fetch]="FETCH_SIZE"
C[write]="WRITE_SIZE"
mkdir -p "$OUT"
for p in $PASSES; do
  rocprofv3 --pmc ${C[$p]} --output-format csv -d "$OUT/$p" -o p -- python3 "$PROBE" > "$OUT/$p.log" 2>&1 || echo "pass $p failed (see $OUT/$p.log)"
done
