#!/bin/bash
# Developer aid: stage-kernel variants on narrow slabs (config 3 at a fixed 16 / 2 stages per iteration), tuning build made on the box.
# usage (through gpurun, from the repo root): bash tools/ab_narrow.sh
set -e
cd "$GRAFT_REPO_ROOT"
cp topolow_amd/csrc/libtopolow_relax.so /tmp/lib_prod.so
make -C topolow_amd/csrc tuning > /tmp/tuning.log 2>&1 || { tail -20 /tmp/tuning.log; exit 1; }
for st in 16 2; do
for v in -1 21 27 25 28; do
  TOPOLOW_SLAB_VARIANT=$v timeout -k 10 300 python bench.py --stages $st --steps 20 --warmup 5 --no-cpu-baseline --no-precision-f64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['by_kind']['stage_kernel']
print('stages $st variant $v: value', round(d['value']), 'stage kernel us', round(k['avg_launch_us'],2), 'frac', round(k['frac'],3), 'launches', k['launches'], 'mae', round(d['final_mae'],5))"
done
done
cp /tmp/lib_prod.so topolow_amd/csrc/libtopolow_relax.so
