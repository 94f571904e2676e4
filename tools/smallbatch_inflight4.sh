#!/bin/bash
# final code: the reference's validation batch (bs 1; image + mirror = bs 2) with 1 / 3 / 4 steps in flight (PipelinedValidation's mechanism; default 4)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sb_inflight4
rm -rf $OUT && mkdir -p $OUT
for b in 1 2; do
  for n in 1 3 4 1 3 4; do
    python3 bench.py --batch $b --inflight $n --steps 120 --warmup 6 --no-cpu-baseline --no-extras > $OUT/b${b}_n$n.json 2>> $OUT/log.txt || { tail -5 $OUT/log.txt; exit 1; }
    echo "batch $b inflight $n $(python3 -c "import json; d=json.loads(open('$OUT/b${b}_n$n.json').read().strip().splitlines()[-1]); print('img/s', d['value'], 'ms/step', d['ms_per_step'], 'host issue ms/step', d['host_issue_ms_per_step'])")" | tee -a $OUT/summary.txt
  done
done
