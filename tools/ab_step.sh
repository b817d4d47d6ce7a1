#!/bin/bash
# Same-box A/B of the denoise step: the round-1 tree (_r1ref/, exported from commit 408900b) against the working tree,
# alternating runs (device-to-device spread between gpurun boxes is +-5 %, larger than most single changes).
# _r1ref/ is git-ignored; recreate it with  mkdir _r1ref && git archive 408900b | tar -x -C _r1ref && make -C _r1ref/mmgt_amd/csrc
# (and take it out of .gpurunignore for the run: it travels to the GPU box with the snapshot).
for i in 1 2; do
  (cd _r1ref && python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('r1  ', round(d['ms_per_step'],2))")
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cur ', round(d['ms_per_step'],2))"
done
