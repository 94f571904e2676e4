#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_* (run on the GPU box through gpurun): kernel trace + stats, then FETCH_SIZE and
# WRITE_SIZE in passes of their own, then two passes of SQ counters (matrix-pipe busy, LDS bank conflicts, instruction
# mix).  Counter passes carry no trace domain but --kernel-trace; the program comes directly after `--`.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
# per-kernel passes run with one batch in flight (kernels of different steps must not overlap in a per-step breakdown);
# the first pass is the DEFAULT command (three batches in flight) and keeps rocprof's own per-kernel statistics of it
B="bench.py --no-cpu-baseline --no-extras --inflight 1"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt_default -- python3 bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 > gpurun_out/prof/kt_default.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt -- python3 $B --steps 3 --warmup 2 > gpurun_out/prof/kt.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_fetch -- python3 $B --steps 2 --warmup 1 > gpurun_out/prof/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_write -- python3 $B --steps 2 --warmup 1 > gpurun_out/prof/pmc_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d gpurun_out/prof/sq_a -- python3 $B --eager --steps 2 --warmup 1 > gpurun_out/prof/sq_a.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/prof/sq_b -- python3 $B --eager --steps 2 --warmup 1 > gpurun_out/prof/sq_b.log 2>&1 || exit 1
tail -n 1 gpurun_out/prof/kt.log | cut -c1-200
