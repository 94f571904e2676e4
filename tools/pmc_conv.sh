#!/bin/bash
# SQ counter passes over one conv shape (run on the GPU box through gpurun; writes gpurun_out/pmc_conv)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_conv
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_conv/a -- python3 tools/run_conv_split.py "$@" > gpurun_out/pmc_conv/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM --output-format csv -d gpurun_out/pmc_conv/b -- python3 tools/run_conv_split.py "$@" > gpurun_out/pmc_conv/b.log 2>&1
ls gpurun_out/pmc_conv/*/*/* | head
