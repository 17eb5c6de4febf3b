set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_g -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_g.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_g_fetch -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_g_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_g_write -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_g_write.log 2>&1
python3 tools/summarize_rocprof.py gpurun_out/prof_g/*/*_kernel_stats.csv > gpurun_out/g_stats.txt
python3 tools/summarize_pmc.py gpurun_out/pmc_g_fetch > gpurun_out/g_fetch.txt
python3 tools/summarize_pmc.py gpurun_out/pmc_g_write > gpurun_out/g_write.txt
tail -1 gpurun_out/prof_g.log | cut -c1-300
