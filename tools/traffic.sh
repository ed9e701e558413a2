#!/bin/bash
# HBM traffic of one kind's kernel for library variants: tools/traffic.sh <kind> <kernel> <name> [<name> ...]
# (two rocprofv3 --pmc passes per variant: FETCH_SIZE x 2 [gfx950: 128-byte requests tallied at 64], WRITE_SIZE; KB per dispatch)
cd "$(dirname "$0")/.." || exit 1
REPO=$PWD; kind=$1; kern=$2; shift 2
export TMPDIR=/tmp
for name in "$@"; do
  if [ "$name" = shipped ]; then unset ZAFX_LIBRARY; else export ZAFX_LIBRARY=$REPO/tools/bin/libzafx_${name}.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/traffic_${name}_$c; rm -rf $d
    (cd /tmp && ZAFX_BENCH_LIVE_TRAFFIC=0 timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- python $REPO/bench.py --kind $kind --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python - "$f" "$kern" "$c" "$name" <<'PY'
import csv, sys
f, kern, c, name = sys.argv[1:]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and kern in r["Kernel_Name"]]
mult = 2.0 if c == "FETCH_SIZE" else 1.0
print(f"{name} {kern} {c}: {sum(v) / max(len(v), 1) * 1024 * mult / 1e9:.4f} GB per launch ({len(v)} dispatches)")
PY
  done
done
