#!/bin/bash
# subsurface path after a change: its tests, the teapot frame's time, when the group kernel's waves end, its cycle accounting
# (bash tools/sss_check.sh, on the GPU box; results under gpurun_out/)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT && mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_render.py -m gpu -x -q --timeout 200 --timeout-method thread -k "subsurface or sss" -s > gpurun_out/sss_tests.log 2>&1
rc=$?; tail -6 gpurun_out/sss_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -u tools/sss_bench.py > gpurun_out/sss_new.log 2>&1 || exit 1
RT_GATHER_SPAN=1 SSS_KS=1000 timeout -k 10 200 python -u tools/sss_bench.py > gpurun_out/sss_span_new.txt 2>&1 || exit 1
RT_GATHER_PROF=1 SSS_KS=1000 timeout -k 10 200 python -u tools/sss_bench.py > gpurun_out/sss_prof_new.txt 2>&1 || exit 1
grep subsurface gpurun_out/sss_new.log; grep "waves in time" gpurun_out/sss_span_new.txt | tail -1; grep -A4 "^packed group" gpurun_out/sss_prof_new.txt | tail -5
