#!/bin/bash
# SQ counters of the bin head's two schedules (OCV_BH_VARIANT=r3|r4) on tools/exp_binhead.py; run on the GPU box through gpurun.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_bh
rm -rf $OUT && mkdir -p $OUT
for v in ${VARIANTS:-r3 r4}; do
  export OCV_BH_VARIANT=$v
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/${v}_a -- python3 tools/exp_binhead.py child > $OUT/${v}_a.log 2>&1 || { tail -3 $OUT/${v}_a.log; exit 1; }
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/${v}_b -- python3 tools/exp_binhead.py child > $OUT/${v}_b.log 2>&1 || { tail -3 $OUT/${v}_b.log; exit 1; }
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/${v}_c -- python3 tools/exp_binhead.py child > $OUT/${v}_c.log 2>&1 || echo "(pass c unavailable for $v)"
done
python3 tools/pmc_binhead_sum.py $OUT
