#!/bin/bash
# SQ counter passes over pointwise shapes (indices of tools/run_pw.py); writes gpurun_out/pmc_pw
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_pw; mkdir -p gpurun_out/pmc_pw
export OCV_PW_CFG=tile
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d gpurun_out/pmc_pw/a -- python3 tools/run_pw.py "$@" > gpurun_out/pmc_pw/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_pw/b -- python3 tools/run_pw.py "$@" > gpurun_out/pmc_pw/b.log 2>&1
ls gpurun_out/pmc_pw/*/*/* | head
