#!/bin/bash
# Same-box A/B of the denoise step with and without a switch of the library's mmgt_tune table (host switches: twin_attention=0,
# oz3=0, rowgemm=0, ...; kernel knobs: splitk=0, ...), alternating runs.   usage: bash tools/ab_env.sh twin_attention=0 [rounds]
for i in $(seq 1 ${2:-3}); do
  MMGT_TUNE="$1" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('with $1 ', round(d['ms_per_step'],2))"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default          ', round(d['ms_per_step'],2))"
done
