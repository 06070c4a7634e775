#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
for a in random32 random; do
echo "== probe ($a)"
timeout 300 python tools/ivf_slow_probe.py --assign $a > gpurun_out/r03_t5_probe_$a.log 2>&1; echo "exit $?"; tail -1 gpurun_out/r03_t5_probe_$a.log
done
