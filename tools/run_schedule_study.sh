#!/bin/bash
# GPU: the schedule's constants (unfolding iterations, k / S bound) against the pinned problems, 64 production-path seeds
# each, on a tuning build made on the box (knobs: csrc/relax_common.h).  usage (through gpurun): bash tools/run_schedule_study.sh
set -e
cd "$GRAFT_REPO_ROOT"
cp topolow_amd/csrc/libtopolow_relax.so /tmp/lib_prod.so
make -C topolow_amd/csrc tuning > /tmp/tuning.log 2>&1 || { tail -20 /tmp/tuning.log; exit 1; }
O=gpurun_out/schedule
mkdir -p $O
: > $O/summary.txt
run() {
  echo "== $1" | tee -a $O/summary.txt
  for P in cfg3 cfg3gen_2048 cfg3gen_1500 cfg3b_1500 syn1500_h3n2params cfg3gen_1500_lowk syn1500_ndim2 syn2000_ndim3_sparse syn7168_ndim2 syn7168_ndim3_sparse; do
    env $2 python tests/study/gpu_contract_study.py $O/$1_$P.json $P 64 2>/dev/null | tee -a $O/summary.txt
  done
}
run base "TL_EARLY_ITERS=16"
run it8 "TL_EARLY_ITERS=8"
run k30 "TL_STAGE_K=3.0"
run it8k30 "TL_EARLY_ITERS=8 TL_STAGE_K=3.0"
run it8k35 "TL_EARLY_ITERS=8 TL_STAGE_K=3.5"
cp /tmp/lib_prod.so topolow_amd/csrc/libtopolow_relax.so
