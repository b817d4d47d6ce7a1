#!/bin/bash
# Same-box A/B of the denoise step: _prev/ (an export of some earlier commit, built in place) against the working tree,
# alternating runs.   usage: bash tools/ab_prev.sh [rounds]
for i in $(seq 1 ${1:-3}); do
  (cd _prev && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', round(d['ms_per_step'],2))")
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cur ', round(d['ms_per_step'],2))"
done
