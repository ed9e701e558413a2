for v in d1p1 d1p2 d2p2 d1p3 d2p0; do for i in 1 2; do
  ZAFX_ISTFT_VARIANT=$v timeout 120 python bench.py --kind istft --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('istft $v', d['ms_per_step'], d['roofline']['frac'])"
done; done
