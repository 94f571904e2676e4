#!/bin/bash
# End-to-end A/B of the pre-split pointwise route (bench.py, one batch in flight, sequential img/s): which layers pay in the real forward
cd $GRAFT_REPO_ROOT
run() { local tag=$1; shift; env "$@" python bench.py --no-extras --no-cpu-baseline --steps 60 --warmup 3 --inflight 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['value'], d['step_latency_ms'])"; }
for rep in 1 2; do
run "off                 " OCV_PW_HL=0
run "on (default policy) " OCV_PW_HL=1
run "expand only         " OCV_PW_HL=1 OCV_PW_HL_PROJECT_MIN_CIN=99999
run "project only (st.5) " OCV_PW_HL=1 OCV_PW_HL_MIN_CIN=99999
run "expand, M<=8000     " OCV_PW_HL=1 OCV_PW_HL_PROJECT_MIN_CIN=99999 OCV_PW_HL_MAX_ROWS=8000
run "expand, Cin>=512    " OCV_PW_HL=1 OCV_PW_HL_PROJECT_MIN_CIN=99999 OCV_PW_HL_MIN_CIN=512
run "expand+proj >=1824  " OCV_PW_HL=1 OCV_PW_HL_PROJECT_MIN_CIN=1824 OCV_PW_HL_WEIGHT_RATIO=0
done
