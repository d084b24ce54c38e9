#!/bin/bash
# round 5, GPU call 1: hardware probe of the f16x3 arithmetic (fp16 subnormal MFMA inputs, the limb split, shader clock),
# first run of csrc/rmlp_h3.hip (parity tests + A/B timing against rmlp.hip), sustained clock / power under the MFMA kernels
cd /root/repo; O=/root/repo/gpurun_out/r5c1; mkdir -p $O
timeout 120 scripts/probes/h3_probe > $O/h3_probe.log 2>&1; echo "probe rc=$?"; cat $O/h3_probe.log | cut -c1-200
timeout 900 python -m pytest tests/test_rmlp_gpu.py -x -q -s > $O/pytest_rmlp.log 2>&1; echo "rmlp tests rc=$?"; tail -15 $O/pytest_rmlp.log | cut -c1-250
timeout 600 python scripts/bench_rmlp.py > $O/bench_rmlp.log 2>&1; echo "bench_rmlp rc=$?"; cat $O/bench_rmlp.log | cut -c1-250
timeout 600 python scripts/sustained_clock.py 8 > $O/sustained_clock.log 2>&1; echo "clock rc=$?"; tail -8 $O/sustained_clock.log | cut -c1-900
cp gpurun_out/r05_sustained_clock.json $O/ 2>/dev/null
ls /sys/class/drm/ > $O/sysfs_ls.txt 2>&1; ls /sys/class/drm/card*/device/ >> $O/sysfs_ls.txt 2>&1; ls /sys/class/drm/card*/device/hwmon/*/ >> $O/sysfs_ls.txt 2>&1
