#!/bin/bash
# same-box comparison of several library builds on four decoder conv shapes: args = variant names (MAIN = product build)
for s in "16 240 320 128 128" "16 240 320 280 128" "16 60 80 1088 512" "16 30 40 1024 1024"; do
  for v in "$@" "$@"; do
    L=objcavit_amd/lib/variants/$v.so; [ "$v" = MAIN ] && L=objcavit_amd/lib/libobjcavit_hip.so
    OCV_ITERS=1500 OCV_LIB_PATH=$L python3 tools/run_conv_split.py $s 2>&1 | grep shape | sed "s/^/  $v: /"
  done
done
