#!/bin/bash
# rocm-smi power / sclk samples while ONE conv shape runs back to back ($1 = "B H W Cin Cout"): is the chip power-capped?
OCV_ITERS=30000 python3 tools/run_conv_split.py $1 > gpurun_out/conv_long.log 2>&1 &
BP=$!
sleep 20
for i in $(seq 1 8); do rocm-smi --showpower --showclocks 2>&1 | grep -i "Power (W)\|sclk" ; sleep 0.3; done
wait $BP
grep shape gpurun_out/conv_long.log
