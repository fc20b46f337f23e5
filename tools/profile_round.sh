#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline object refers to (run on the GPU box through gpurun):
#   pass 1: --kernel-trace --stats   -> per-kernel durations
#   pass 2..: --pmc (one counter group per pass, with --kernel-trace only)  -> HBM bytes, MFMA busy, waits
# Output: gpurun_out/prof/<pass>/... (csv) ; summarise with tools/summarize_profile.py into profiles/.
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-device-loop --no-extras --no-timesplit"   # the driver's timed region (20 steps, 5 warm-up) without the CPU leg, the fallback leg and the extra brackets
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- $CMD > "$OUT/trace.log" 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc$i" -o bench -- $CMD > "$OUT/pmc$i.log" 2>&1
done
cd "$ROOT"
python tools/summarize_profile.py "$OUT" gpurun_out/prof_summary
# the raw per-launch csv files are tens of MB: only the summaries travel back
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -delete
