#!/bin/bash
# A/B of OCV_HEAD_OVERLAP (heads' conv3x3 beside the token chain) on one box: sequential (--inflight 1) and the default 3 in flight.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/head_overlap
rm -rf $OUT && mkdir -p $OUT
for b in ${BATCHES:-16 1}; do
  for v in 0 1 0 1; do
    OCV_HEAD_OVERLAP=$v python3 bench.py --batch $b --inflight 1 --steps 40 --warmup 3 --no-cpu-baseline --no-extras > $OUT/seq_b${b}_v$v.json 2>> $OUT/log.txt || { tail -5 $OUT/log.txt; exit 1; }
    echo "bs $b sequential OCV_HEAD_OVERLAP=$v: $(python3 -c "import json,sys; d=json.loads(open('$OUT/seq_b${b}_v$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
  done
done
for v in 0 1 0 1; do
  OCV_HEAD_OVERLAP=$v python3 bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras > $OUT/pipe_v$v.json 2>> $OUT/log.txt || { tail -5 $OUT/log.txt; exit 1; }
  echo "bs 16, 3 in flight OCV_HEAD_OVERLAP=$v: $(python3 -c "import json,sys; d=json.loads(open('$OUT/pipe_v$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['value_sequential'])")"
done
