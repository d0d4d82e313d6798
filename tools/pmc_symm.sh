#!/bin/bash
# Developer aid: builds tools/symm_probe.hip on the GPU box, runs it, and collects three rocprofv3 counter passes
# (--pmc only) over it.  usage (from the repo root, through gpurun): bash tools/pmc_symm.sh
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
hipcc -O3 -std=c++17 --offload-arch=gfx950 -I topolow_amd/csrc -o /tmp/symm_probe tools/symm_probe.hip
timeout -k 5 120 /tmp/symm_probe 10000 0 50 > gpurun_out/symm_probe.log 2>&1
run() { timeout -k 5 120 rocprofv3 --pmc $2 --output-format csv -d gpurun_out/symm_pmc_$1 -- /tmp/symm_probe 10000 0 3 > gpurun_out/symm_pmc_$1.log 2>&1; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY"
run b "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"
run c "SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES"
run d "GRBM_GUI_ACTIVE GRBM_COUNT"
