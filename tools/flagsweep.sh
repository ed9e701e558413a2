for kind in mel mfcc cqt; do
  for name in shipped wprio bias0 trk nopost relax; do
    if [ "$name" = shipped ]; then unset ZAFX_LIBRARY; else export ZAFX_LIBRARY=$PWD/tools/bin/libzafx_${name}.so; fi
    ms=$(python bench.py --kind "$kind" --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['parity'].get('within_tolerance'))")
    echo "$kind $name: $ms"
  done
done
