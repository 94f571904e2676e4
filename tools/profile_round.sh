#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_* (run on the GPU box through gpurun): kernel trace + stats, then FETCH_SIZE and
# WRITE_SIZE in passes of their own (counter passes carry no other trace domain).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt -- python3 bench.py --steps 3 --warmup 2 > gpurun_out/prof/kt.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 > gpurun_out/prof/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_write -- python3 bench.py --steps 2 --warmup 1 > gpurun_out/prof/pmc_write.log 2>&1 || exit 1
tail -n 1 gpurun_out/prof/kt.log | cut -c1-200
