# One workload's profiles: kernel trace + stats, then one counter pass per run (HBM reads, HBM writes, SQ).
#   bash tools/profile_round.sh TAG WORKLOAD     (on the GPU box; results under gpurun_out/)
set -e
TAG=${1:-a}; WL=${2:-c3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-cpu-baseline --no-variants --workload $WL"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- $B --steps 20 --warmup 5 > gpurun_out/prof_$TAG.log 2>&1
python3 tools/summarize_rocprof.py gpurun_out/prof_$TAG/*/*_kernel_stats.csv > gpurun_out/${TAG}_stats.txt
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_fetch -- $B --steps 4 --warmup 2 > gpurun_out/pmc_${TAG}_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_write -- $B --steps 4 --warmup 2 > gpurun_out/pmc_${TAG}_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${TAG}_sq -- $B --steps 4 --warmup 2 > gpurun_out/pmc_${TAG}_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_${TAG}_lds -- $B --steps 4 --warmup 2 > gpurun_out/pmc_${TAG}_lds.log 2>&1
for k in fetch write sq lds; do python3 tools/summarize_pmc.py gpurun_out/pmc_${TAG}_$k > gpurun_out/${TAG}_$k.txt; done
python3 tools/make_traffic_json.py $WL gpurun_out/pmc_${TAG}_fetch gpurun_out/pmc_${TAG}_write gpurun_out/pmc_${TAG}_sq gpurun_out/pmc_${TAG}_lds > gpurun_out/${TAG}_traffic_$WL.json
tail -1 gpurun_out/prof_$TAG.log | cut -c1-400
