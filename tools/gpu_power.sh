#!/bin/bash
# Energy per instruction class at the package power limit (tools/ubench/power_probe.hip) and the shader clock / power of
# unet_kernel back to back, sampled with rocm-smi.  Usage (GPU box): bash tools/gpu_power.sh > gpurun_out/power.txt
sample() { /opt/rocm/bin/rocm-smi --showclocks --showpower 2>&1 | awk '/sclk/{c=$NF} /Power \(W\)/{p=$NF} END{printf "sclk %s power %s W", c, p}'; }
echo "idle: $(sample)"
for cls in mfma16 mfma32 lds valu l2 idle; do
  ./build_tmp/power_probe $cls 5 > /tmp/pp_$cls.txt 2>&1 &
  PID=$!
  sleep 2.5; s1=$(sample); sleep 0.5; s2=$(sample); sleep 0.5; s3=$(sample)
  wait $PID
  echo "$(cat /tmp/pp_$cls.txt)"
  echo "    $s1 | $s2 | $s3"
done
for n in 512 1024 2048; do
  REPS=$((3000000 / n * 10)) timeout 100 python tools/unet_forward_loop.py $n > /tmp/loop_$n.txt 2>&1 &
  PID=$!
  sleep 7; s1=$(sample); sleep 0.4; s2=$(sample)
  wait $PID
  echo "unet_kernel back to back, $(grep 'n=' /tmp/loop_$n.txt | cut -c1-120)"
  echo "    $s1 | $s2"
done
