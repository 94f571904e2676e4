#!/bin/bash
# A/B of OCV_SKIP_OVERLAP (the decoder's skip-part convolutions beside the encoder: modules/DenseFeatureExtractor.py SkipPrepass) on
# one box, one batch at a time (--inflight 1; with several batches in flight the switch is off by default like every fork).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/skip_overlap
rm -rf $OUT && mkdir -p $OUT
for b in ${BATCHES:-16 1 2}; do
  for v in 0 1 0 1; do
    OCV_SKIP_OVERLAP=$v python3 bench.py --batch $b --inflight 1 --steps 40 --warmup 3 --no-cpu-baseline --no-extras > $OUT/seq_b${b}_v$v.json 2>> $OUT/log.txt || { tail -5 $OUT/log.txt; exit 1; }
    echo "bs $b one at a time OCV_SKIP_OVERLAP=$v: $(python3 -c "import json,sys; d=json.loads(open('$OUT/seq_b${b}_v$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")" | tee -a $OUT/summary.txt
  done
done
