#!/bin/bash
# rocm-smi power / sclk samples while bench.py runs with 1 and with 3 batches in flight (is the whole step at the power cap?)
mkdir -p gpurun_out/power
for n in 1 3; do
  python3 bench.py --steps 1500 --warmup 3 --inflight $n --no-cpu-baseline --no-extras > gpurun_out/power/b_$n.log 2>&1 &
  BP=$!
  sleep 25
  for i in $(seq 1 25); do rocm-smi --showpower --showclocks 2>&1 | grep -i "power (W)\|sclk" ; sleep 0.2; done > gpurun_out/power/smi_$n.txt
  wait $BP
  echo "inflight=$n: $(cut -c80-170 gpurun_out/power/b_$n.log | tail -1)"
  grep -i "power" gpurun_out/power/smi_$n.txt | awk '{print $NF}' | sort -n | awk '{a[NR]=$1} END {print "  power W: min",a[1],"median",a[int(NR/2)+1],"max",a[NR], "n",NR}'
  grep -i "sclk" gpurun_out/power/smi_$n.txt | grep -o "([0-9]*Mhz)" | tr -d '()Mhz' | sort -n | awk '{a[NR]=$1} END {print "  sclk MHz: min",a[1],"median",a[int(NR/2)+1],"max",a[NR]}'
done
