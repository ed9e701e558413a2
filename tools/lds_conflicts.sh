#!/bin/bash
# LDS bank-conflict share of the dominant kernel of each transform:  gpurun -- 'bash tools/lds_conflicts.sh "mdct imdct istft"'
cd /tmp; export TMPDIR=/tmp
for k in ${1:-mdct imdct istft}; do
  rm -rf /tmp/pm_$k
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/pm_$k -o $k -- \
      python $GRAFT_REPO_ROOT/bench.py --kind $k --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python - "$k" <<PY
import csv, glob, sys
k = sys.argv[1]
f = glob.glob(f"/tmp/pm_{k}/**/*counter_collection.csv", recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    if "zafx::" in r["Kernel_Name"]:
        acc.setdefault((r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
names = sorted({n for n, _ in acc})
for n in names:
    g = {c: sum(v) / len(v) for (nn, c), v in acc.items() if nn == n}
    print(k, n, {c: round(v / 1e6, 2) for c, v in g.items()}, "conflict/active", round(g.get("SQ_LDS_BANK_CONFLICT", 0) / max(g.get("SQ_LDS_IDX_ACTIVE", 1), 1), 4))
PY
done
