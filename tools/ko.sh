#!/bin/bash
# knock-out variants of one kind: tools/ko.sh <kind> <name> ...  -> kernel_ms per variant (one round; results of the variants are wrong by design)
cd "$(dirname "$0")/.." || exit 1
kind=$1; shift
for name in "$@"; do
  if [ "$name" = shipped ]; then unset ZAFX_LIBRARY; else export ZAFX_LIBRARY=$PWD/tools/bin/libzafx_${name}.so; fi
  ms=$(python bench.py --kind "$kind" --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'])")
  echo "$kind $name: $ms"
done
