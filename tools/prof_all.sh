set -u
cd $GRAFT_REPO_ROOT
timeout -k 10 400 bash profiles/run_profile.sh r02f > gpurun_out/prof_r02f.log 2>&1 || { echo profile failed; exit 1; }
timeout -k 10 300 bash profiles/run_pmc.sh c3f python3 $GRAFT_REPO_ROOT/tools/pmc_run.py c3 > gpurun_out/pmc_c3f.log 2>&1 || { echo pmc c3 failed; exit 1; }
timeout -k 10 200 python bench.py > gpurun_out/bench_r02f.json 2> gpurun_out/bench_r02f.err || { echo bench failed; exit 1; }
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r02f_20.json 2> gpurun_out/bench_r02f_20.err || { echo bench20 failed; exit 1; }
tail -c 600 gpurun_out/bench_r02f.json
