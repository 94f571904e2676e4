#!/bin/bash
# OCV_TAP_SKIP=1 restricted to the last stage (24 skip channels = one padded block): three in flight, with the per-launch table
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/tap_skip32
rm -rf $OUT && mkdir -p $OUT
for v in 0 1 0 1; do
  OCV_TAP_SKIP=$v OCV_TAP_SKIP_MAX_CP=32 python3 bench.py --steps 40 --warmup 3 --no-cpu-baseline > $OUT/pipe_v$v.json 2>> $OUT/log.txt || { tail -5 $OUT/log.txt; exit 1; }
  echo "bs 16, 3 in flight OCV_TAP_SKIP=$v (<= 32 channels): $(python3 -c "import json,sys; d=json.loads(open('$OUT/pipe_v$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['value_sequential'], [(c['shape'], c['ms']) for c in d['convs'] if c['form'].startswith('tap') or '24->128' in c['shape']])")" | tee -a $OUT/summary.txt
done
