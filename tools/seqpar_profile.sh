#!/usr/bin/env bash
# rocprofv3 --kernel-trace of RANK 0 of a 2-rank sequence-parallel forward + backward (tools/seqpar_worker.py): which kernels run, and are
# any ATen elementwise kernels of activation size among them?   bash tools/seqpar_profile.sh <out file> <mode> [dynchunk]
set -uo pipefail
OUTF="$1"; MODE="$2"; DC="${3:-}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
PORT=$((20000 + RANDOM % 20000))
export WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT SMX_MODE="$MODE" SMX_DYNCHUNK="$DC" HSA_ENABLE_IPC_MODE_LEGACY=0
RANK=1 python "$ROOT/tools/seqpar_worker.py" > /tmp/seqpar_r1.log 2>&1 &
P1=$!
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_sp && RANK=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_sp -o p -- python "$ROOT/tools/seqpar_worker.py" > /tmp/seqpar_r0.log 2>&1)
wait $P1
DB=$(find /tmp/prof_sp -name '*.db' | head -1)
mkdir -p "$(dirname "$OUTF")"
{ echo "# rocprofv3 --kernel-trace --stats -- python tools/seqpar_worker.py   (rank 0 of 2; $(tail -1 /tmp/seqpar_r0.log))"
  python - "$DB" <<'PY'
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
gx = "grid_x" if "grid_x" in cols else "grid_size_x"
rows = con.execute(f"select name, count(*), sum(end-start), max({gx}) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
act = 4 * 4096 * 256                                        # elements of one activation tensor of this rank
print(f"total kernel time {tot/1e6:.2f} ms; one activation tensor of this rank = {act} elements")
print(f"{'%':>6} {'calls':>6} {'max threads':>12}  kernel")
aten_big = []
for n, c, t, g in rows:
    short = re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", re.sub(r"smx::", "", n)))[:110]
    print(f"{100*t/tot:6.1f} {c:6d} {g:12d}  {short}")
    if "at::native" in n and g * 4 >= act // 4:              # (a vectorised elementwise kernel handles 4+ elements per thread)
        aten_big.append((short, c, g))
print()
print("ATen kernels with >= 1/16 of an activation tensor's elements in threads (3 steps x 2 layers; the halo concatenation / slices of the conv module's")
print("extended sequence and the gradient zero-fills - the summaries' boundary arithmetic has none):")
for a in aten_big: print("   ", a)
if not aten_big: print("    NONE")
PY
} > "$OUTF"
tail -3 "$OUTF"
