#!/bin/bash
# Samples rocm-smi (power, sclk) every ~0.2 s while a long bench run is in flight: is the chip power-capped under the conv?
rocm-smi --showpower --showclocks --showmaxpower > gpurun_out/smi_idle.txt 2>&1
python3 bench.py --steps 2500 --warmup 3 > gpurun_out/b_long.log 2>&1 &
BP=$!
sleep 30
for i in $(seq 1 12); do rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk\|fclk" ; sleep 0.3; done > gpurun_out/smi_busy.txt
wait $BP
tail -c 600 gpurun_out/b_long.log | head -c 300
