#!/bin/bash
# Same-box A/B of the tail-tile policies at the batch sizes where experts sit just above one 256-row tile.
# usage: tools/ab_tails.sh  (prints tokens, variant, ms per step)
cd "$GRAFT_REPO_ROOT"
for T in 3929 4096 6144 8192; do
  for V in "default" "SGLK_AUX_PRIO=low" "SGLK_AUX_PRIO=high" "SGLK_TAIL_SPLIT=2" "SGLK_TAIL_SPLIT=0"; do
    if [ "$V" = "default" ]; then E=""; else E="$V"; fi
    ms=$(env $E python bench.py --tokens $T --steps 50 --warmup 10 --no-cpu-baseline --no-a8 --no-verify 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])")
    echo "$T $V $ms"
  done
done
