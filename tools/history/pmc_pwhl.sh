#!/bin/bash
# SQ / traffic counter passes over one pre-split pointwise layer: tools/pmc_pwhl.sh TAG M Cin Cout act   -> gpurun_out/pmc_pwhl/TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
d=gpurun_out/pmc_pwhl/$tag
rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $d/a -- python3 tools/run_pwhl_one.py "$@" > $d/a.log 2>&1 || { tail -3 $d/a.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $d/b -- python3 tools/run_pwhl_one.py "$@" > $d/b.log 2>&1 || { tail -3 $d/b.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_IFETCH --output-format csv -d $d/c -- python3 tools/run_pwhl_one.py "$@" > $d/c.log 2>&1 || { tail -3 $d/c.log; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $d/f -- python3 tools/run_pwhl_one.py "$@" > $d/f.log 2>&1 || { tail -3 $d/f.log; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $d/w -- python3 tools/run_pwhl_one.py "$@" > $d/w.log 2>&1 || { tail -3 $d/w.log; }
python3 tools/pmc_pwhl_summary.py $d
