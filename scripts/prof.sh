#!/bin/bash
# rocprofv3 kernel stats of the default bench command; summary csv -> gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats; mkdir -p /root/repo/gpurun_out/prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/prof/bench_under_rocprof.json 2> /tmp/prof_stats.err
find /tmp/prof_stats -name "*stats*" -exec cp {} /root/repo/gpurun_out/prof/ \;
ls -la /root/repo/gpurun_out/prof/
head -12 /root/repo/gpurun_out/prof/*kernel_stats.csv
