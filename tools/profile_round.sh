set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_h -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_h.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_h_fetch -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_h_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_h_write -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_h_write.log 2>&1
python3 tools/summarize_rocprof.py gpurun_out/prof_h/*/*_kernel_stats.csv > gpurun_out/h_stats.txt
python3 tools/summarize_pmc.py gpurun_out/pmc_h_fetch > gpurun_out/h_fetch.txt
python3 tools/summarize_pmc.py gpurun_out/pmc_h_write > gpurun_out/h_write.txt
tail -1 gpurun_out/prof_h.log | cut -c1-300
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_h_sq -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_h_sq.log 2>&1
python3 tools/summarize_pmc.py gpurun_out/pmc_h_sq > gpurun_out/h_sq.txt
