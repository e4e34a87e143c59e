#!/usr/bin/env bash
# HBM traffic + L2 hit rate + SQ wait breakdown of the grouped wgrad kernel (separate --pmc passes, kernel-trace only).
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp; export TMPDIR=/tmp
export WHICH="${1:-layer}"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  d=/tmp/pmcw_$(echo $c | tr ' ' '_'); rm -rf $d
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- python3 "$ROOT/tools/one_wgroup.py" 64000 $WHICH > /dev/null 2>&1 || true
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for fn in glob.glob("/tmp/pmcw_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "wgrad_group" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
f = acc.get("FETCH_SIZE"); w = acc.get("WRITE_SIZE")
if f and w:
    f, w = sum(f)/len(f)*1024, sum(w)/len(w)*1024
    import os
    label = "wgrad_group bf16 (8 weights of a C2b layer) over 64000 frames [algorithmic 1021.8 MB: 983.0 operands + 5.8 dW + 33 slab set]"
    if os.environ.get("WHICH") == "c2a":
        label = "wgrad_group bf16 (8 weights of a C2a layer, d_model 512) over 64000 frames [operands 2031.6 MB + 23.1 MB of dW]"
    print(f"{label}: "
          f"launches={len(acc['FETCH_SIZE'])} FETCH_SIZE={f/1e6:.1f} MB (x2 corrected {2*f/1e6:.1f} MB)  WRITE_SIZE={w/1e6:.1f} MB  "
          f"traffic(corrected)={(2*f+w)/1e6:.1f} MB per launch")
PY
