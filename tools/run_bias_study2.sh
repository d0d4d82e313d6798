#!/bin/bash
# GPU study: does a finer stage split while the layout unfolds change the basin statistics? (config 3)
set -e
O=gpurun_out/bias2
mkdir -p $O
python tests/study/gpu_schedule_bias.py $O/cfg3_e0.json cfg3 8 trace:0 trace:64 >> $O/log.txt 2>&1
for E in 12:16 12:64 40:16 40:64 100:32; do
  TOPOLOW_SLAB_EARLY=$E python tests/study/gpu_schedule_bias.py $O/cfg3_e$E.json cfg3 8 trace:0 >> $O/log.txt 2>&1
done
cat $O/log.txt
