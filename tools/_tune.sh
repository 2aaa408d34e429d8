for w in P4 S8 PVR4 PVR8spx; do
  echo "== $w"
  SVR_TUNE_DEBUG=1 python bench.py --workload $w --no-cpu-baseline --steps 4 2>&1 | grep -E "^\[tune\]|^\{" | cut -c1-200
done
