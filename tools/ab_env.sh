#!/bin/bash
# Same-box A/B of the denoise step with and without an environment switch of the host code (MMGT_NO_OZ3=1, MMGT_NO_ROWGEMM=1, ...),
# alternating runs.   usage: bash tools/ab_env.sh MMGT_NO_OZ3=1 [rounds]
for i in $(seq 1 ${2:-3}); do
  env "$1" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('with $1 ', round(d['ms_per_step'],2))"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default          ', round(d['ms_per_step'],2))"
done
