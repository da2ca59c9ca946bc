#!/bin/bash
# Same-box A/B of the kernel choice between the mid-kernel regime and the 256-row regime (72 .. 256 rows per expert).
cd "$GRAFT_REPO_ROOT"
for T in 1536 2048 2560 3072; do
  for V in "default" "SGLK_MOE_TILE_M=128" "SGLK_MOE_TILE_M=96" "SGLK_MOE_TILE_M=256"; do
    if [ "$V" = "default" ]; then E=""; else E="$V"; fi
    ms=$(env $E python bench.py --tokens $T --steps 50 --warmup 10 --no-cpu-baseline --no-a8 --no-verify 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['stage_ms'])")
    echo "$T | $V | $ms"
  done
done
