#!/bin/bash
for cfg in "0 0" "1 0" "2 0" "3 0" "0 1" "1 1" "2 1"; do
  set -- $cfg
  echo "dbg=$1 S_cap=$2"
  W2VS_TN8=1 W2VS_TN8_DBG=$1 W2VS_TN8_S=$2 timeout -k 10 120 python tools/wgrad_group_probe.py 6544 || exit 1
done
