cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pg && rocprofv3 --kernel-trace -d /tmp/pg -o p -- python /root/repo/bench.py --config c2a --batch 10 --frames 375 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > /tmp/pg.log 2>&1
DB=$(find /tmp/pg -name '*.db' | head -1); python /root/repo/tools/prof_gaps.py $DB 15
rm -rf /tmp/pg && rocprofv3 --kernel-trace -d /tmp/pg -o p -- python /root/repo/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-extra-points > /tmp/pg.log 2>&1
DB=$(find /tmp/pg -name '*.db' | head -1); python /root/repo/tools/prof_gaps.py $DB 6
