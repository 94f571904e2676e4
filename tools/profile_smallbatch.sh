#!/bin/bash
# The batches the reference's own validation loop runs (main.py:58 forces bs 1; GraphBinsLM.py:159,173 = image + mirror):
# bench.py --batch b --inflight 1 (one batch after the other) for b in 1 2 8 16, and a rocprofv3 kernel trace of the same
# command per batch.  Run on the GPU box through gpurun; condense with `python tools/make_smallbatch_profile.py <tag>`.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/smallbatch${TAG:+_$TAG}
rm -rf $OUT && mkdir -p $OUT
for b in ${BATCHES:-1 2 8 16}; do
  python3 bench.py --batch $b --inflight 1 --steps 40 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bs$b.json 2> $OUT/bs$b.log || { tail -5 $OUT/bs$b.log; exit 1; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_bs$b -- python3 bench.py --batch $b --inflight 1 --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $OUT/kt_bs$b.log 2>&1 || { tail -5 $OUT/kt_bs$b.log; exit 1; }
  echo "bs $b done: $(cut -c1-200 $OUT/bs$b.json)"
done
