#!/bin/bash
# rocprofv3 PMC counters for an arbitrary python script (one counter group per pass; never combined with sys/hip traces).
# usage: tools/pmc_cmd.sh <outdir> <script.py> [args...]
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OUT/pass$i" -- python3 "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done <<GROUPS
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_WAVES
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS
GROUPS
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.csv" -size +2M -delete
cat "$OUT/summary.txt"
