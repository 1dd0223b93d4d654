#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout-s> '<command>'  -- retries while the pod's GPU slots are busy (exit code 3 / "transient")
for k in $(seq 1 30); do
  out=$(timeout 5000 gpurun --timeout "$1" -- "$2" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 60; continue; fi
  echo "$out"; exit 0
done
echo "no GPU slot after 30 tries"; exit 3
