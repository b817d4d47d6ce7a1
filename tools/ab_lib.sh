#!/bin/bash
# Same-box A/B of an experiment build of the library (MMGT_LIB=path) against the product library on the denoise step, alternating runs:
#   bash tools/ab_lib.sh mmgt_amd/csrc/build/libmmgt_hip_wt.so        (ROUNDS=3)
R=${ROUNDS:-3}
for i in $(seq 1 $R); do
  MMGT_LIB="$1" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-calib 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2))"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-calib 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product', round(d['ms_per_step'],2))"
done
