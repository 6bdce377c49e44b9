#!/bin/bash
# PMC passes over the training step: MFMA busy cycles + GRBM_GUI_ACTIVE (busy fraction and shader clock per kernel), waits.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_train"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/scripts/train_bench.py 8 96 2"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace.log" 2>&1; echo "rc=$?"
timeout 900 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/pmc1" -o pmc1 -- $CMD > "$OUT/pmc1.log" 2>&1; echo "rc=$?"
timeout 900 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace -d "$OUT/pmc2" -o pmc2 -- $CMD > "$OUT/pmc2.log" 2>&1; echo "rc=$?"
python "$REPO/scripts/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1; head -120 "$OUT/summary.txt"
find "$OUT" -type f \( -name "*.db" -o -name "*.pftrace" -o -name "*.json" \) -size +6M -delete
