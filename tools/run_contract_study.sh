#!/bin/bash
# GPU: 64 production-path seeds per pinned problem against the committed oracle distributions
# (tests/study/gpu_contract_study.py); the summary lines go to profiles/r02_contract_study.txt.
set -e
O=gpurun_out/contract
mkdir -p $O
: > $O/summary.txt
for P in cfg3 cfg3gen_2048 cfg3gen_1500 cfg3b_1500 syn1500_h3n2params cfg3gen_1500_lowk cfg3gen_1500_eps1e-6 cfg3gen_1500_eps1e-10 syn1500_ndim2 syn2000_ndim3_sparse; do
  python tests/study/gpu_contract_study.py $O/$P.json $P 64 2>/dev/null | tee -a $O/summary.txt
done
