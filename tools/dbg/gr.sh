#!/bin/bash
# usage: tools/dbg/gr.sh <timeout-seconds> '<command>'   -- gpurun with retries while every GPU slot of the pod is busy
for i in $(seq 1 30); do
  out=$(/usr/local/graft/bin/gpurun --timeout "$1" -- "$2" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 45; continue; fi
  echo "$out"; exit 0
done
echo "$out"; exit 3
