#!/bin/bash
# Same-box A/B of the denoise step: $PREV (an export of some earlier commit, built in place: `mkdir .ab_prev && git archive <commit> | tar -x -C .ab_prev &&
# make -C .ab_prev/mmgt_amd/csrc`; kept out of the tree between uses -- remove it after the gpurun call) against the working tree, alternating runs.
#   usage: bash tools/ab_prev.sh [rounds]
PREV=${PREV:-.ab_prev}
for i in $(seq 1 ${1:-3}); do
  (cd $PREV && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-calib 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', round(d['ms_per_step'],2))")
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-calib 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cur ', round(d['ms_per_step'],2))"
done
