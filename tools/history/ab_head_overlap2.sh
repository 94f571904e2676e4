#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/head_overlap2
rm -rf $OUT && mkdir -p $OUT
run() {  # label, env...
  label=$1; shift
  env "$@" python3 bench.py --batch $B --inflight $INF --steps 30 --warmup 3 --no-cpu-baseline --no-extras > $OUT/$label.json 2>> $OUT/log.txt || { tail -5 $OUT/log.txt; exit 1; }
  echo "bs $B inflight $INF $label: $(python3 -c "import json,sys; d=json.loads(open('$OUT/$label.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
}
for B in 16 1; do
INF=1
run obj1_head1 OCV_OBJ_OVERLAP=1 OCV_HEAD_OVERLAP=1
run obj0_head1_tok0 OCV_HEAD_OVERLAP=1 OCV_TOKEN_OVERLAP=0
run obj0_head0_tok1 OCV_HEAD_OVERLAP=0 OCV_TOKEN_OVERLAP=1
run obj1_head1 OCV_OBJ_OVERLAP=1 OCV_HEAD_OVERLAP=1
run obj0_head1_tok0 OCV_HEAD_OVERLAP=1 OCV_TOKEN_OVERLAP=0
run obj0_head0_tok1 OCV_HEAD_OVERLAP=0 OCV_TOKEN_OVERLAP=1
done
B=16; INF=3
run obj1_head1 OCV_OBJ_OVERLAP=1 OCV_HEAD_OVERLAP=1
run obj0_head1_tok0 OCV_HEAD_OVERLAP=1 OCV_TOKEN_OVERLAP=0
run obj0_head0_tok1 OCV_HEAD_OVERLAP=0 OCV_TOKEN_OVERLAP=1
run obj1_head1 OCV_OBJ_OVERLAP=1 OCV_HEAD_OVERLAP=1
run obj0_head1_tok0 OCV_HEAD_OVERLAP=1 OCV_TOKEN_OVERLAP=0
run obj0_head0_tok1 OCV_HEAD_OVERLAP=0 OCV_TOKEN_OVERLAP=1
