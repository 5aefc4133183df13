#!/bin/bash
R0=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/p_ro -- python $R0/bench.py --workload rollout --steps 512 --warmup 256 > /tmp/b_ro.json 2>/dev/null
cut -c1-200 /tmp/b_ro.json
f=$(find /tmp/p_ro -name "*.db" | head -1)
if [ -n "$f" ]; then ROWS=14 timeout 60 python $R0/tools/rocprof_summary.py /tmp/s_ro.md "rollout" $f < /dev/null | head -22 | cut -c1-150; fi
