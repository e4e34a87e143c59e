cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_c5 && rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -o p -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
DB=$(find /tmp/prof_c5 -name '*.db' | head -1)
cd $GRAFT_REPO_ROOT; python tools/prof_summary.py $DB 4 | head -24
