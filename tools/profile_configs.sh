#!/bin/bash
# Every BASELINE configuration through bench.py: the JSON line of `bench.py --config c` (pipelined,
# sequential, sustained and exact-fp32 rates, kernels / convs tables, roofline, cpu_baseline) and a rocprofv3 kernel trace of
# the same workload with one batch in flight (per-step breakdown by launch grid).  Run on the GPU box through gpurun;
# condense with `python tools/make_configs_profile.py r03`.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/configs
for c in ${CONFIGS:-1 2 3 4}; do
  python3 bench.py --config $c --steps 20 --warmup 3 --cpu-budget 25 > gpurun_out/configs/cfg$c.json 2> gpurun_out/configs/cfg$c.log || { tail -5 gpurun_out/configs/cfg$c.log; exit 1; }
  rm -rf gpurun_out/prof_cfg$c && mkdir -p gpurun_out/prof_cfg$c
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg$c/kt -- python3 bench.py --config $c --no-cpu-baseline --no-extras --inflight 1 --steps 3 --warmup 2 > gpurun_out/prof_cfg$c/kt.log 2>&1 || { tail -5 gpurun_out/prof_cfg$c/kt.log; exit 1; }
  echo "config $c done: $(cut -c1-160 gpurun_out/configs/cfg$c.json)"
done
