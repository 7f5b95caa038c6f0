#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc_sweep.json
# (run on the GPU box from the repo root:  bash tools/profile_round.sh r01c).
# Kernel trace and every PMC group are SEPARATE runs of the same command (the counters do
# not fit one pass, and --pmc must not be combined with other trace domains).
tag=${1:-r01x}
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check"
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o sweep -- $B > $out/trace.log 2>&1; echo "trace rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU \
    --output-format csv -d $out/pmc_sq -o sweep -- $B > $out/pmc_sq.log 2>&1; echo "pmc_sq rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_fetch -o sweep -- $B > $out/pmc_fetch.log 2>&1; echo "pmc_fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT TCC_MISS --output-format csv -d $out/pmc_write -o sweep -- $B > $out/pmc_write.log 2>&1; echo "pmc_write rc=$?"
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 \
    --output-format csv -d $out/pmc_lds -o sweep -- $B > $out/pmc_lds.log 2>&1; echo "pmc_lds rc=$?"
python3 tools/summarize_profile.py $tag
