for k in stft stft1; do for i in 1 2 3; do
  timeout 120 python bench.py --kind $k --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$k', d['ms_per_step'], d['roofline']['frac'], d.get('max_rel_err_vs_numpy'))"
done; done
