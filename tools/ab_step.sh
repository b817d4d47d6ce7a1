#!/bin/bash
# Same-box A/B of the denoise step: the round-1 tree (_r1ref/, exported from commit 408900b) against the working tree,
# alternating runs (device-to-device spread between gpurun boxes is +-5 %, larger than most single changes).
for i in 1 2; do
  (cd _r1ref && python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('r1  ', round(d['ms_per_step'],2))")
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cur ', round(d['ms_per_step'],2))"
done
