#!/bin/bash
# A/B of the transformer tails' few-token form (feed-forward chunks of a row block over G workgroups, in-launch combine): OCV_TAIL_GROUPS=1
# (one workgroup per row block, round 3) against the automatic choice and forced G, one batch at a time, one box.
cd $GRAFT_REPO_ROOT
run() { # label env batch extra
  env $2 python3 bench.py $4 --batch $3 --inflight 1 --steps 60 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/ab_x.json 2>> gpurun_out/ab_log.txt || { tail -5 gpurun_out/ab_log.txt; exit 1; }
  echo "$4 bs $3 $1: $(python3 -c "import json; d=json.loads(open('gpurun_out/ab_x.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
}
for i in 1 2; do
  for b in 1 2; do
    run "G=1" OCV_TAIL_GROUPS=1 $b ""; run "auto" OCV_X=0 $b ""; run "G=4" OCV_TAIL_GROUPS=4 $b ""
  done
  run "G=1" OCV_TAIL_GROUPS=1 8 ""; run "auto" OCV_X=0 8 ""; run "G=2" OCV_TAIL_GROUPS=2 8 ""
  run "G=1" OCV_TAIL_GROUPS=1 16 ""; run "G=2" OCV_TAIL_GROUPS=2 16 ""
  run "G=1" OCV_TAIL_GROUPS=1 8 "--config 3"; run "auto" OCV_X=0 8 "--config 3"; run "G=4" OCV_TAIL_GROUPS=4 8 "--config 3"
done
