#!/bin/bash
# sample sclk / power while a kernel loop runs:  clock_watch.sh "<command>"
( eval "$1" > /tmp/cw_cmd.log 2>&1 ) &
PID=$!
sleep 2.0
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' '; echo
  sleep 0.5
done
wait $PID
tail -3 /tmp/cw_cmd.log
