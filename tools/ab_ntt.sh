#!/bin/bash
# A/B of the NTT pass variants on the GPU box
for cpt in 1 0; do for map in 0 1; do
  echo "== ZKHIP_NTT_CPT=$cpt ZKHIP_NTT_MAP=$map"
  ZKHIP_NTT_CPT=$cpt ZKHIP_NTT_MAP=$map python tools/time_ops.py 2>&1 | grep -E "ntt_pass|coset_lde|dft"
done; done
