#!/bin/bash
set -e
O=gpurun_out/bias4
mkdir -p $O
for P in cfg3 cfg3gen_2048 cfg3gen_1500 syn1500_h3n2params cfg3b_1500; do
  NS=32
  for E in 16:64 16:32 16:16 32:64; do
    TOPOLOW_SLAB_EARLY=$E python tests/study/gpu_relabel_study.py $O/${P}_e$E.json $P $NS 0 >> $O/log.txt 2>&1
  done
  python tests/study/gpu_relabel_study.py $O/${P}_S16.json $P $NS 16 >> $O/log.txt 2>&1
done
python tests/study/gpu_relabel_study.py $O/cfg3_gs.json cfg3 6 gs >> $O/log.txt 2>&1
python tests/study/gpu_relabel_study.py $O/cfg3gen_2048_gs.json cfg3gen_2048 20 gs >> $O/log.txt 2>&1
python tests/study/gpu_relabel_study.py $O/cfg3gen_1500_gs.json cfg3gen_1500 20 gs >> $O/log.txt 2>&1
cat $O/log.txt
