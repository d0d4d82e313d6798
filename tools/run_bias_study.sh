#!/bin/bash
# GPU study driver: final-MAE statistics of the device schedules (tests/study/gpu_schedule_bias.py).
set -e
O=gpurun_out/bias
mkdir -p $O
for P in cfg3gen_1500 syn1500_h3n2params cfg3gen_2048; do
  for L in 0 1 2; do
    TOPOLOW_SLAB_DAMP=$L python tests/study/gpu_schedule_bias.py $O/${P}_d$L.json $P 20 slab:0 slab:8 slab:16 slab:64 >> $O/log.txt 2>&1
  done
  python tests/study/gpu_schedule_bias.py $O/${P}_gs.json $P 20 gs >> $O/log.txt 2>&1
done
for L in 0 1 2; do
  TOPOLOW_SLAB_DAMP=$L python tests/study/gpu_schedule_bias.py $O/cfg3_d$L.json cfg3 8 slab:0 slab:8 slab:16 slab:32 >> $O/log.txt 2>&1
done
python tests/study/gpu_schedule_bias.py $O/cfg3_trace.json cfg3 4 trace:0 gs >> $O/log.txt 2>&1
cat $O/log.txt
