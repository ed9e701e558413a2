#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line of every transform, rocprofv3 kernel statistics of each, the two
# HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) and -- for the kernels with matrix-core work -- a pass of
# the SQ counters (MFMA busy cycles, MFMA instructions, LDS instructions / bank conflicts).  Everything lands under
# gpurun_out/; tools/summarize_profiles.py turns it into the files committed under profiles/.
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh'
cd "$(dirname "$0")/.." || exit 1
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
KINDS=${KINDS:-"stft istft mdct imdct mel mfcc cqt dct stft_offgrid stft4096 stft4096_h1024 istft4096 mdct4096"}   # KINDS="stft" re-collects the headline only
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_all.json" 2> "$OUT/bench_all.log"   # the driver's own command: every config in one line
for k in $KINDS; do
  timeout 300 python bench.py --kind $k --no-cpu-baseline > "$OUT/bench_$k.json" 2> "$OUT/bench_$k.log"
done
timeout 300 python bench.py --kind stft --layout TF --no-cpu-baseline > "$OUT/bench_stft_tf.json" 2>> "$OUT/bench_stft.log"
cd /tmp || exit 1
for k in $KINDS; do
  rm -rf "$OUT/prof_$k"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$k" -o $k -- \
      python "$REPO/bench.py" --kind $k --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/prof_$k.log" 2>&1   # (enough timed launches that the placement probes' launches -- same kernel, other buffers -- move the mean by < 2 %)
done
for k in $KINDS; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$OUT/pmc_${k}_$c"
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_${k}_$c" -o $k -- \
        python "$REPO/bench.py" --kind $k --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_${k}_$c.log" 2>&1
  done
done
for k in mel mfcc dct cqt stft; do
  rm -rf "$OUT/pmc_${k}_SQ"
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS \
      --kernel-trace --output-format csv -d "$OUT/pmc_${k}_SQ" -o $k -- \
      python "$REPO/bench.py" --kind $k --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_${k}_SQ.log" 2>&1
  rm -rf "$OUT/pmc_${k}_LDS"
  timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d "$OUT/pmc_${k}_LDS" -o $k -- \
      python "$REPO/bench.py" --kind $k --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_${k}_LDS.log" 2>&1
done
find "$OUT" -name "*.csv" -size +8M -delete   # per-dispatch traces of the big runs are not needed
ls "$OUT"
