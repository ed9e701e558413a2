#!/bin/bash
# Runs on the GPU box (through gpurun).  ONE process gives the driver-style line AND the rocprofv3 kernel trace / statistics,
# so that profiles/ and the line describe the same placement of the same buffers (round 3: line from one process 0.735, stats
# from another 0.664).  Then a plain (unprofiled) run of the same command -- its line carries the HBM traffic of every config,
# measured by the two --pmc child runs bench.py starts itself -- and the SQ-counter passes of the compute-bound kernels.
# Everything lands under gpurun_out/; tools/summarize_profiles.py turns it into the files committed under profiles/.
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh'
cd "$(dirname "$0")/.." || exit 1
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
SQ_KINDS=${SQ_KINDS:-"mel mfcc mel_mfcc cqt stft mel64 cqt64 istft8192 imdct8192"}
if [ -z "$SQ_ONLY" ]; then
cd /tmp || exit 1
rm -rf "$OUT/prof_all"
ZAFX_BENCH_INNER_LOG="$OUT/prof_all_launches.json" timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_all" -o all -- \
    python "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_all_profiled.json" 2> "$OUT/bench_all_profiled.log"
cp "$OUT/bench_detail.json" "$OUT/bench_detail_profiled.json" 2>/dev/null
cd "$REPO" || exit 1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_all.json" 2> "$OUT/bench_all.log"   # the driver's own command
cp "$OUT/bench_detail.json" "$OUT/bench_detail_plain.json" 2>/dev/null   # (the counter passes below write their own)
fi   # (SQ_ONLY=1: only the counter passes below, for the kinds in SQ_KINDS)
cd /tmp || exit 1
for k in $SQ_KINDS; do
  rm -rf "$OUT/pmc_${k}_SQ" "$OUT/pmc_${k}_LDS"
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS \
      --kernel-trace --output-format csv -d "$OUT/pmc_${k}_SQ" -o $k -- \
      python "$REPO/bench.py" --kind $k --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_${k}_SQ.log" 2>&1
  timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
      --kernel-trace --output-format csv -d "$OUT/pmc_${k}_LDS" -o $k -- \
      python "$REPO/bench.py" --kind $k --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_${k}_LDS.log" 2>&1
done
find "$OUT" -name "*.csv" -size +16M -delete
ls "$OUT"
