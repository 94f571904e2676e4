#!/bin/bash
# A/B of OCV_TAP_SKIP (the skip part of the decoder's last three first-convolutions formed inside the tap-interpolation launch vs as
# a convolution launch of its own): default command (three in flight) and one batch at a time, alternating on one box
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/tap_skip
rm -rf $OUT && mkdir -p $OUT
for v in 0 1 0 1; do
  OCV_TAP_SKIP=$v python3 bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras > $OUT/pipe_v$v.json 2>> $OUT/log.txt || { tail -5 $OUT/log.txt; exit 1; }
  echo "bs 16, 3 in flight OCV_TAP_SKIP=$v: $(python3 -c "import json,sys; d=json.loads(open('$OUT/pipe_v$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['value_sequential'], [(c['shape'], c['ms']) for c in d['convs'] if c['form'].startswith('tap')])")" | tee -a $OUT/summary.txt
done
for b in ${BATCHES:-16 1}; do
  for v in 0 1 0 1; do
    OCV_TAP_SKIP=$v python3 bench.py --batch $b --inflight 1 --steps 40 --warmup 3 --no-cpu-baseline --no-extras > $OUT/seq_b${b}_v$v.json 2>> $OUT/log.txt || { tail -5 $OUT/log.txt; exit 1; }
    echo "bs $b one at a time OCV_TAP_SKIP=$v: $(python3 -c "import json,sys; d=json.loads(open('$OUT/seq_b${b}_v$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")" | tee -a $OUT/summary.txt
  done
done
