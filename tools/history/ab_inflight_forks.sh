#!/bin/bash
# bs 16: slots in flight x forks inside each captured forward (all four side-stream switches forced on / left off), alternating on one box
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/inflight_forks
rm -rf $OUT && mkdir -p $OUT
for rep in 1 2; do
  for cfg in "3 0" "2 1" "2 0" "3 1"; do
    set -- $cfg
    OCV_OBJ_OVERLAP=$2 OCV_TOKEN_OVERLAP=$2 OCV_HEAD_OVERLAP=$2 OCV_SKIP_OVERLAP=$2 python3 bench.py --inflight $1 --steps 40 --warmup 3 --no-cpu-baseline --no-extras > $OUT/n$1_f$2.json 2>> $OUT/log.txt || { tail -5 $OUT/log.txt; exit 1; }
    echo "bs 16, $1 in flight, forks=$2: $(python3 -c "import json,sys; d=json.loads(open('$OUT/n$1_f$2.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")" | tee -a $OUT/summary.txt
  done
done
