#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/fit_pmc1 gpurun_out/fit_pmc2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/fit_pmc1 -o fit -- python3 tools/fit_trace.py > gpurun_out/fit_pmc1.log 2>&1
python3 tools/fit_pmc.py gpurun_out/fit_pmc1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/fit_pmc2 -o fit -- python3 tools/fit_trace.py > gpurun_out/fit_pmc2.log 2>&1
python3 tools/fit_pmc.py gpurun_out/fit_pmc2
