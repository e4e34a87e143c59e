#!/usr/bin/env bash
# HBM traffic per launch (TCC FETCH_SIZE x2-corrected on gfx950 + WRITE_SIZE, separate --pmc passes) of the front-end kernels at
# B = 128 x 20 s, per kernel name, from one run of tools/frontend_bench.py each.
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcfe_$c; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcfe_$c -o p -- python3 "$ROOT/tools/frontend_bench.py" > /dev/null 2>&1 || true
done
python3 - <<'PY'
import csv, glob, collections, re
acc = {c: collections.defaultdict(list) for c in ("FETCH_SIZE", "WRITE_SIZE")}
for c in acc:
    for fn in glob.glob(f"/tmp/pmcfe_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == c:
                n = re.sub(r"^void |smx::|\(.*\)$", "", r["Kernel_Name"])
                acc[c][n].append(float(r["Counter_Value"]))
print("# HBM traffic per launch, front-end kernels at B = 128 x 20 s (FETCH_SIZE KiB x 1024 x 2 [gfx950 correction], WRITE_SIZE KiB x 1024)")
for n in sorted(acc["FETCH_SIZE"], key=lambda k: -sum(acc["FETCH_SIZE"][k])):
    f = acc["FETCH_SIZE"][n]; w = acc["WRITE_SIZE"].get(n, [0.0])
    fm, wm = 2 * 1024 * sum(f) / len(f) / 1e6, 1024 * sum(w) / len(w) / 1e6
    if fm + wm > 5:
        print(f"{n[:88]:88s} launches={len(f):3d}  read {fm:8.1f} MB  write {wm:8.1f} MB")
PY
