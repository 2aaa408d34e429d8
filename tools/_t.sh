for o in "coeff_table=1" "coeff_table=1 fwd_mode=2"; do
  python tools/shard_one.py PVR4 0 1 10 $o 2>&1 | tail -1 | cut -c1-200
done
python tools/shard_one.py PVR8spx 0 1 6 coeff_table=1 2>&1 | tail -1 | cut -c1-200
python -m pytest tests/test_parity_gpu.py tests/test_pvr.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
