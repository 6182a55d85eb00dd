#!/bin/bash
for dbg in 0 1 2 3; do
  echo "== ZKHIP_NTT_DEBUG=$dbg (1 = no loads, 2 = no stores, 3 = neither)"
  ZKHIP_NTT_DEBUG=$dbg python tools/time_ops.py 2>&1 | grep -E "ntt_pass"
done
