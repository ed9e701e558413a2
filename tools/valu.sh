#!/bin/bash
# Instruction counters of one kind's kernel for library variants: tools/valu.sh <kind> <kernel> <name> [<name> ...]
# (one rocprofv3 --pmc pass per variant: SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_INSTS_LDS, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES; mean per dispatch)
cd "$(dirname "$0")/.." || exit 1
REPO=$PWD; kind=$1; kern=$2; shift 2
export TMPDIR=/tmp
for name in "$@"; do
  if [ "$name" = shipped ]; then unset ZAFX_LIBRARY; else export ZAFX_LIBRARY=$REPO/tools/bin/libzafx_${name}.so; fi
  d=/tmp/valu_${name}; rm -rf $d
  (cd /tmp && ZAFX_BENCH_LIVE_TRAFFIC=0 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $d -o p -- python $REPO/bench.py --kind $kind --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python - "$f" "$kern" "$name" <<'PY'
import collections, csv, sys
f, kern, name = sys.argv[1:]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Kernel_Name"].startswith(kern + "<") or r["Kernel_Name"].startswith("void zafx::" + kern + "<") or r["Kernel_Name"] == kern:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(name, kern, " ".join(f"{c}={sum(v) / max(len(v), 1) / 1e6:.2f}M" for c, v in sorted(acc.items())), f"({max((len(v) for v in acc.values()), default=0)} dispatches)")
PY
done
