#!/bin/bash
# Counter passes of tools/placement_pmc.py (slow against fast placement of config 2's spectrum): translation, the L2's write
# requests by destination, and their stalls.  On the GPU box:  bash tools/placement_pmc.sh  ->  gpurun_out/placement_pmc/
cd "$(dirname "$0")/.." || exit 1
REPO=$PWD
OUT=$REPO/gpurun_out/placement_pmc
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp || exit 1
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_REQUEST_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_LEVEL_sum GRBM_UTCL2_BUSY" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum"; do
  i=$((i + 1))
  rm -rf "$OUT/p$i"
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -o p -- python "$REPO/tools/placement_pmc.py" 12 > "$OUT/p$i.log" 2>&1
  tail -2 "$OUT/p$i.log"
  python - "$OUT/p$i" <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))
if not f:
    print("no counter csv"); sys.exit()
rows = [r for r in csv.DictReader(open(f[-1])) if "k_stft_ft16" in r["Kernel_Name"]]
by = collections.defaultdict(dict)
for r in rows:
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)[-12:]
names = sorted({n for d in ids for n in by[d]})
for label, sel in (("slow", ids[:6]), ("fast", ids[6:])):
    print(label, {n: round(sum(by[d].get(n, 0) for d in sel) / len(sel)) for n in names})
PY
done
find "$OUT" -name "*.csv" -size +4M -delete
