#!/bin/bash
# r06 (review item 3): where the cycles of the fp32 headline kernel go, from counters instead of timelines.  The stand-alone bench of
# conv3x3_wf4.h on the dominant instantiation / shape (relu-on-load, no residual, 64 -> 64 @ 12 x 544 x 992) under rocprofv3 PMC passes
# (SQ block: 8 counters per pass; --pmc never together with other trace domains).  Run on the GPU box:  bash scripts/wf4_pmc.sh
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/wf4_pmc"
mkdir -p "$OUT"
BIN="$REPO/scripts/probes/wf4_bench"
[ -x "$BIN" ] || { echo "build scripts/probes/wf4_bench first (see its header)"; exit 1; }
export WF4_SHAPES="${WF4_SHAPES:-12,544,992,64,64,1,0}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_avail.txt" 2>&1
grep -o "SQ_[A-Z0-9_]*" "$OUT/counters_avail.txt" | sort -u > "$OUT/sq_counters.txt"
pass() {   # name, counters...   (ONLY="p6": just that pass)
  n=$1; shift
  [ -n "$ONLY" ] && [ "$ONLY" != "$n" ] && return
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$n" -o "$n" -- "$BIN" > "$OUT/$n.log" 2>&1; echo "$n rc=$?"
}
pass p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
pass p2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT_LDS_ONLY
pass p3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INSTS_VALU_MFMA_MOPS_F32
pass p4 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
pass p5 SQ_WAVES_EQ_64 SQ_WAIT_INST_ANY SQ_IFETCH SQ_ITEMS SQ_INSTS_WAVE32_LDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_EXP_GDS
pass p6 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_THREAD_CYCLES_VALU
python "$REPO/scripts/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
grep -A12 "conv3x3_wf4_kernel" "$OUT/summary.txt" | head -120
find "$OUT" -type f \( -name "*.db" -o -name "*.pftrace" -o -name "*.json" \) -size +6M -delete
