#!/bin/bash
# GPU: two-stage iterations as symmetric half sweeps against the row-owner stages (TOPOLOW_SYMMETRIC_TWO_STAGE=0) on the
# pinned problems the symmetric sweep applies to (>= 7168 points), 64 production-path seeds each, and config 3b (10 %
# censored).  usage (through gpurun): bash tools/run_two_stage_study.sh
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/two_stage
mkdir -p $O
: > $O/summary.txt
for V in 1 0; do
  echo "== TOPOLOW_SYMMETRIC_TWO_STAGE=$V" | tee -a $O/summary.txt
  for P in cfg3 syn7168_ndim2 syn7168_ndim3_sparse; do
    TOPOLOW_SYMMETRIC_TWO_STAGE=$V python tests/study/gpu_contract_study.py $O/v${V}_$P.json $P 64 2>/dev/null | tee -a $O/summary.txt
  done
done
