#!/bin/bash
# GPU-box sweep of the digest Q x M kernel's ring depth (multi_sweep: 32768 candidates, 1..16 queries).
for cfg in "MSC_DIGEST_SLOTS=2" "MSC_DIGEST_SLOTS=3" "MSC_DIGEST_SLOTS=4" "MSC_DIGEST_SLOTS=6" "MSC_DIGEST_SLOTS=8"; do
  echo "== $cfg"
  env $cfg timeout -k 10 300 python tools/multi_sweep.py 32768 2>&1 | tail -3 || exit 1
done
