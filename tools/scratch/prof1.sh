cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_x && rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
DB=$(find /tmp/prof_x -name '*.db' | head -1)
cd $GRAFT_REPO_ROOT; python tools/prof_summary.py $DB 7 | head -40
