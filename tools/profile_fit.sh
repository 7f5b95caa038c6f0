#!/bin/bash
# rocprofv3 passes over the fit path (tools/fit_nll_prof.py: gpUtils._nll-style evaluations at N = 512 / 1152 / 3072 / 4096):
#   bash tools/profile_fit.sh r05      (GPU box, repo root)  ->  gpurun_out/prof_fit_<tag>/{kernel_stats.csv, fit_pmc.json}
# Kernel trace and each PMC group are SEPARATE runs (--pmc is never combined with other trace domains).
tag=${1:-r05}
out=gpurun_out/prof_fit_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 tools/fit_nll_prof.py"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o fit -- $B > $out/trace.log 2>&1; echo "trace rc=$?"
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU \
    --output-format csv -d $out/pmc1 -o fit -- $B > $out/pmc1.log 2>&1; echo "pmc1 rc=$?"
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS \
    --output-format csv -d $out/pmc2 -o fit -- $B > $out/pmc2.log 2>&1; echo "pmc2 rc=$?"
python3 tools/fit_pmc_json.py $out
