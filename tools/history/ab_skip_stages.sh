#!/bin/bash
# how many of the decoder's skip-part convolutions ride the side stream (OCV_SKIP_STAGES: 1 = 240x320 only, forked behind encoder
# stage 2; 2 = + 120x160, behind stage 3; 3 = + 60x80, behind stage 4), one batch at a time, alternating on one box
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/skip_stages
rm -rf $OUT && mkdir -p $OUT
for b in ${BATCHES:-16 1}; do
  for v in 3 2 1 3 2 1; do
    OCV_SKIP_STAGES=$v python3 bench.py --batch $b --inflight 1 --steps 40 --warmup 3 --no-cpu-baseline --no-extras > $OUT/seq_b${b}_v$v.json 2>> $OUT/log.txt || { tail -5 $OUT/log.txt; exit 1; }
    echo "bs $b one at a time OCV_SKIP_STAGES=$v: $(python3 -c "import json,sys; d=json.loads(open('$OUT/seq_b${b}_v$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")" | tee -a $OUT/summary.txt
  done
done
