#!/bin/bash
# Samples power / clocks with rocm-smi while a command runs.  usage: clock_probe.sh OUT -- cmd...
out=$1; shift; shift
"$@" > "$out.cmd.log" 2>&1 &
pid=$!
sleep 2.5   # past the import / setup phase
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature" >> "$out"
  echo "--" >> "$out"
  sleep 0.7
done
wait $pid
