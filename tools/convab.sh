for s in "16 60 80 1088 512" "16 240 320 128 128" "16 30 40 2224 1024" "16 120 160 552 256" "16 240 320 280 128"; do
  OCV_LIB_PATH=$PWD/objcavit_amd/lib/variants/$1.so python3 tools/run_conv_split.py $s 2>&1 | grep shape | sed "s/^/   $1: /"
  OCV_LIB_PATH=$PWD/objcavit_amd/lib/variants/$2.so python3 tools/run_conv_split.py $s 2>&1 | grep shape | sed "s/^/   $2: /"
done
