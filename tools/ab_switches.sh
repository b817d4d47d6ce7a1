#!/bin/bash
# Same-box A/B of MMGT_TUNE switches on the denoise step, each against the defaults, alternating runs:
#   bash tools/ab_switches.sh "attn_nomax=0" "rconv=0" ...        (ROUNDS=3)
R=${ROUNDS:-3}
for sw in "$@"; do
  for i in $(seq 1 $R); do
    MMGT_TUNE="$sw" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-calib 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$sw', round(d['ms_per_step'],2))"
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-calib 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', round(d['ms_per_step'],2))"
  done
done
