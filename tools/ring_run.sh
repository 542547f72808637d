#!/bin/bash
# GPU-box sweep of the Q x M kernel variants (TQ x ring slots), plus SQ counters of the default one.
mkdir -p gpurun_out/sweep
for cfg in "MSC_MULTI_NO_RING=1 MSC_MULTI_TQ=4" "MSC_MULTI_TQ=4 MSC_RING_SLOTS=2" "MSC_MULTI_TQ=4 MSC_RING_SLOTS=3" "MSC_MULTI_TQ=4 MSC_RING_SLOTS=4" "MSC_MULTI_TQ=8 MSC_RING_SLOTS=2" "MSC_MULTI_TQ=8 MSC_RING_SLOTS=3" "MSC_MULTI_TQ=8 MSC_RING_SLOTS=4"; do
  echo "== $cfg"
  env $cfg timeout -k 10 300 python tools/multi_sweep.py 32768 2>&1 | tail -3 || exit 1
done
