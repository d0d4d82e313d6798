#!/bin/bash
# GPU: how fine must the unfolding phase be?  Stages per iteration during the first TL_EARLY_ITERS iterations (production:
# 16 stages, 8 iterations) against the pinned problems, 64 production-path seeds each, on a tuning build made on the
# box (knobs: csrc/relax_common.h).  usage (through gpurun): bash tools/run_early_stage_study.sh
set -e
cd "$GRAFT_REPO_ROOT"
cp topolow_amd/csrc/libtopolow_relax.so /tmp/lib_prod.so
make -C topolow_amd/csrc tuning > /tmp/tuning.log 2>&1 || { tail -20 /tmp/tuning.log; exit 1; }
O=gpurun_out/early
mkdir -p $O
: > $O/summary.txt
run() {
  echo "== $1" | tee -a $O/summary.txt
  for P in cfg3gen_1500 cfg3b_1500 cfg3gen_1500_eps1e-6 cfg3gen_2048 syn2000_ndim3_sparse cfg3; do
    env $2 python tests/study/gpu_contract_study.py $O/$1_$P.json $P 64 2>/dev/null | tee -a $O/summary.txt
  done
}
run s16 "TL_EARLY_STAGES=16"
run s32 "TL_EARLY_STAGES=32"
run s64 "TL_EARLY_STAGES=64"
run s64it16 "TL_EARLY_STAGES=64 TL_EARLY_ITERS=16"
cp /tmp/lib_prod.so topolow_amd/csrc/libtopolow_relax.so
