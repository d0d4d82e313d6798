#!/bin/bash
# The rocprofv3 passes behind profiles/r03_*: kernel trace + stats, then one --pmc pass per counter
# group (never combined with traces).  Run on the GPU box from the repo root:
#   bash tools/profile_round3.sh gpurun_out/prof_r3
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=$1
rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_kt -- python3 bench.py --no-cpu-baseline --min-timed 0.05 > ${P}_kt.log 2>&1
# counter passes: the same command, shortened.  The adaptive policy mixes 16-stage (unfolding), 2-stage and
# 1-stage launches of the same kernel; tools/summarize_profiles.py separates the one-stage launches (the
# 4 N^2-byte sweeps the per-launch figures are quoted for) by their duration.
rocprofv3 --pmc FETCH_SIZE --output-format csv -d ${P}_fetch -- python3 bench.py --min-timed 0 --no-cpu-baseline > ${P}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d ${P}_write -- python3 bench.py --min-timed 0 --no-cpu-baseline > ${P}_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d ${P}_sq -- python3 bench.py --min-timed 0 --no-cpu-baseline > ${P}_sq.log 2>&1
echo done
