#!/bin/bash
# Same-box A/B of the denoise step under two MMGT_TUNE settings of the CURRENT library, alternating runs.
#   usage: bash tools/ab_tune.sh "g16_pb=1" "" [rounds]        ("" = the defaults)
A="$1"; B="$2"
for i in $(seq 1 ${3:-3}); do
  MMGT_TUNE="$A" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('A [$A]', round(d['ms_per_step'],2))"
  MMGT_TUNE="$B" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B [$B]', round(d['ms_per_step'],2))"
done
