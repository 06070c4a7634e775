#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/sweep_prepass.py --rows 21250000 --steps 30 --ladders "256,16;2048,256,16;1024,128,16;512,64,16;128,16;1024,64,16;256,64,16;512,128,32;1024,256,64,16;2048,512,128,32;64,16" > gpurun_out/sweep21b.log 2>&1; echo "sweep exit $?"; grep ladder gpurun_out/sweep21b.log | tail -40
