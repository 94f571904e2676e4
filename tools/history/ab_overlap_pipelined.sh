#!/bin/bash
# One box, alternating: the default bench (3 batches in flight) under the side-stream switches.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_overlap
rm -rf $OUT && mkdir -p $OUT
for i in 1 2 3; do
  for cfg in "default OCV_X=0" "nofork OCV_OBJ_OVERLAP=0 OCV_TOKEN_OVERLAP=0 OCV_HEAD_OVERLAP=0" "tokfork OCV_OBJ_OVERLAP=0 OCV_TOKEN_OVERLAP=1 OCV_HEAD_OVERLAP=0" "objfork_head OCV_OBJ_OVERLAP=1 OCV_HEAD_OVERLAP=1"; do
    set -- $cfg; label=$1; shift
    env "$@" python3 bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras > $OUT/${label}_$i.json 2>> $OUT/log.txt || { tail -5 $OUT/log.txt; exit 1; }
    echo "$label run $i: $(python3 -c "import json; d=json.loads(open('$OUT/${label}_$i.json').read().strip().splitlines()[-1]); print(d['value'], d['value_sequential'])")"
  done
done
