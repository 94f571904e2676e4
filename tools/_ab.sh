#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for cfg in "1536 64" "1824 76" "3072 128"; do
    set -- $cfg
    for b in 16 1; do
      OCV_SE_FUSED_MAXC=$1 OCV_SE_FUSED_MAXR=$2 python3 bench.py --batch $b --inflight 1 --steps 40 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/ab_x.json 2>> gpurun_out/ab_log.txt || { tail -5 gpurun_out/ab_log.txt; exit 1; }
      echo "maxC $1 maxR $2 bs $b run $i: $(python3 -c "import json; d=json.loads(open('gpurun_out/ab_x.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
    done
  done
done
