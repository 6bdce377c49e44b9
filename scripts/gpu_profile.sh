#!/bin/bash
# rocprofv3 passes over the bench command (run on the GPU box via gpurun).  Summaries land in
# gpurun_out/prof_*; copy what should be judged into profiles/.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
PREC="${PREC:-fp32}"
OTHERS="${OTHERS:-fp32d,bf16x3,f16f8,mixed}"     # every engine in one run: one pmc_traffic.json for all kernel names; OTHERS=none: the engine
                                                      # alone (-> profiles/pmc_traffic_<engine>.json: `mixed` shares kernel names with f16f8 / fp16)
OUT="$REPO/gpurun_out/prof_${PREC}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps ${STEPS:-2} --warmup 1 --precision $PREC --others $OTHERS --no-cpu-baseline --no-roofline --no-parity --no-flow --no-train"
echo "== kernel trace + stats"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace.log" 2>&1; echo "rc=$?"
echo "== pmc pass 1 (SQ / MFMA busy / vector-ALU instructions)"
timeout 900 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace -d "$OUT/pmc1" -o pmc1 -- $CMD > "$OUT/pmc1.log" 2>&1; echo "rc=$?"
echo "== pmc pass 2 (LDS / waits)"
timeout 900 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace -d "$OUT/pmc2" -o pmc2 -- $CMD > "$OUT/pmc2.log" 2>&1; echo "rc=$?"
echo "== pmc pass 3 (HBM read)"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc3" -o pmc3 -- $CMD > "$OUT/pmc3.log" 2>&1; echo "rc=$?"
echo "== pmc pass 4 (HBM write)"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc4" -o pmc4 -- $CMD > "$OUT/pmc4.log" 2>&1; echo "rc=$?"
cd "$OUT" && find . -type f | head -50 && du -sh .
python "$REPO/scripts/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1; cat "$OUT/summary.txt" | head -90
# the merged gpurun_out/ is capped at 64 MiB: drop the big per-dispatch databases after summarising
find "$OUT" -type f \( -name "*.db" -o -name "*.pftrace" -o -name "*.json" \) -size +6M -delete
