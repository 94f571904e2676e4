#!/bin/bash
# One box, alternating: bench.py at bs 16 with 1 - 5 batches in flight.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_inflight
rm -rf $OUT && mkdir -p $OUT
for i in 1 2; do
  for n in ${INFLIGHT:-3 2 4 5 1}; do
    python3 bench.py --inflight $n --steps 48 --warmup 3 --no-cpu-baseline --no-extras > $OUT/n${n}_$i.json 2>> $OUT/log.txt || { tail -5 $OUT/log.txt; exit 1; }
    echo "inflight $n run $i: $(python3 -c "import json; d=json.loads(open('$OUT/n${n}_$i.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
  done
done
