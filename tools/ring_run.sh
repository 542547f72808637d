timeout 600 python -m pytest tests -q -m gpu -k "multi" -x 2>&1 | tail -5
for cfg in "MSC_MULTI_NO_RING=1 MSC_MULTI_TQ=4" "MSC_MULTI_TQ=4 MSC_RING_SLOTS=2" "MSC_MULTI_TQ=4 MSC_RING_SLOTS=3" "MSC_MULTI_TQ=4 MSC_RING_SLOTS=4" "MSC_MULTI_NO_RING=1 MSC_MULTI_TQ=8" "MSC_MULTI_TQ=8 MSC_RING_SLOTS=2" "MSC_MULTI_TQ=8 MSC_RING_SLOTS=3" "MSC_MULTI_TQ=8 MSC_RING_SLOTS=4"; do
  echo "== $cfg"
  env $cfg timeout 300 python tools/multi_sweep.py 32768 2>&1 | tail -2
done
