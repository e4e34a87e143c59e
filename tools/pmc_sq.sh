#!/usr/bin/env bash
# SQ busy/stall counters of one GEMM shape (one counter group per pass, kernel-trace only).
# usage: pmc_sq.sh LAYOUT N K M [epi]
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp; export TMPDIR=/tmp
GROUPS_=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_ANY")
i=0
for g in "${GROUPS_[@]}"; do
  rm -rf /tmp/sq_$i; rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/sq_$i -o p -- python3 "$ROOT/tools/one_gemm.py" "$@" > /tmp/sq_$i.log 2>&1 || tail -3 /tmp/sq_$i.log
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for fn in glob.glob("/tmp/sq_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "gemm_" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
