#!/bin/bash
for cfg in "2 1 off" "2 4 off" "2 4 on"; do
  echo "== $cfg"; timeout 600 python tools/repeat_probe_synth.py $cfg 2>&1 | grep "^call" | cut -c1-420
done
