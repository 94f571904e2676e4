#!/bin/bash
# A/B of two library builds over the eight decoder conv shapes, same process order, same box: $1, $2 = variant names
for s in "16 240 320 128 128" "16 240 320 280 128" "16 120 160 552 256" "16 120 160 256 256" "16 60 80 1088 512" "16 60 80 512 512" "16 30 40 2224 1024" "16 30 40 1024 1024"; do
  for v in $1 $2 $1 $2; do
    L=objcavit_amd/lib/variants/$v.so; [ "$v" = MAIN ] && L=objcavit_amd/lib/libobjcavit_hip.so
    OCV_ITERS=1500 OCV_LIB_PATH=$L python3 tools/run_conv_split.py $s 2>&1 | grep shape | sed "s/^/  $v: /"
  done
done
