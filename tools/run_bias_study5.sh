#!/bin/bash
# GPU study: floor of the adaptive stage count after the unfolding phase (1, 2 or 4 stages per iteration)
set -e
O=gpurun_out/bias5
mkdir -p $O
for M in 4 2 1; do
  for P in cfg3 cfg3gen_2048 cfg3gen_1500 syn1500_h3n2params cfg3b_1500; do
    TOPOLOW_MIN_STAGES=$M python tests/study/gpu_minstage_study.py $O/${P}_m$M.json $P 32 >> $O/log.txt 2>&1
  done
  TOPOLOW_MIN_STAGES=$M python bench.py --no-cpu-baseline > $O/bench_m$M.json 2>> $O/log.txt || echo "bench failed m=$M" >> $O/log.txt
done
cat $O/log.txt
python - <<'PY'
import json
for m in (4, 2, 1):
    try:
        b = json.load(open(f"gpurun_out/bias5/bench_m{m}.json"))
        print("min_stages", m, "value", round(b["value"]), "it/s; stage kernel", round(b["roofline"]["avg_launch_us"], 1), "us, frac",
              round(b["roofline"]["frac"], 3), "stages/iter", b["config"]["stages_per_iteration"], "whole run", round(b["whole_run"]["iterations_per_s"]),
              "final", b["whole_run"]["final_mae"])
    except Exception as e:
        print("min_stages", m, "no bench:", e)
PY
