#!/bin/bash
for cfg in "1 item-side 1 empty_shortcut" "2 item-side 1 empty_shortcut" "2 item-side 2 empty_shortcut" "2 item-side 1 full" "2 gather-both 0 empty_shortcut"; do
  echo "== $cfg"; timeout 300 python tools/repeat_probe.py $cfg 2>&1 | grep "^rank 0" | cut -c1-330
done
