#!/bin/bash
# A/B of library variants on ONE box: tools/ab.sh <kind> <name> [<name> ...]   (names of tools/bin/libzafx_<name>.so; "shipped" = the in-tree library)
# three alternating rounds of `bench.py --kind <kind>`; prints kernel_ms per variant and round
cd "$(dirname "$0")/.." || exit 1
kind=$1; shift
for round in 1 2 3; do
  for name in "$@"; do
    if [ "$name" = shipped ]; then unset ZAFX_LIBRARY; else export ZAFX_LIBRARY=$PWD/tools/bin/libzafx_${name}.so; fi
    ms=$(python bench.py --kind "$kind" --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['parity'].get('within_tolerance'))")
    echo "$kind $name round $round: $ms"
  done
done
