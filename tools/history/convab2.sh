for s in "16 60 80 1088 512" "16 240 320 128 128" "16 30 40 2224 1024" "16 120 160 552 256" "16 240 320 280 128" "16 120 160 256 256" "16 30 40 1024 1024" "16 60 80 512 512"; do
  OCV_CONV_ONESHOT=1 python3 tools/run_conv_split.py $s 2>&1 | grep shape | sed "s/^/   oneshot: /"
  python3 tools/run_conv_split.py $s 2>&1 | grep shape | sed "s/^/   persist: /"
done
