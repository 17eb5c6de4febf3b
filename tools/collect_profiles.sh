# Copies the summaries of a profile run from gpurun_out/ into profiles/ (tracked), named per round.
#   bash tools/collect_profiles.sh r02 r02b      (round prefix, run tag under gpurun_out/)
set -e
R=$1; T=$2
hdr() { echo "# $1"; }
{ hdr "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-variants --workload c3 --steps 20 --warmup 5 (C3: 1M Gaussians, 1920x1080; tools/profile_round.sh)"; cat gpurun_out/${T}_stats.txt; } > profiles/${R}_bench_c3_kernel_stats.txt
{ hdr "the same for --workload c3w (the C3 cloud with every opacity 0.1: no early termination)"; cat gpurun_out/${T}_c3w_stats.txt; } > profiles/${R}_bench_c3w_kernel_stats.txt
for WL in c5 c2; do { hdr "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-variants --workload $WL --steps 10 --warmup 3 (tools/profile_workloads.sh)"; cat gpurun_out/${T}_${WL}_stats.txt; } > profiles/${R}_bench_${WL}_kernel_stats.txt; done
if [ -f gpurun_out/unfused_stats.txt ]; then { hdr "rocprofv3 --kernel-trace --stats -- python3 bench.py --unfused ... (C3 through the reference's operator API: PyTorch activations feeding GaussianRasterizer; tools/profile_unfused.sh)"; cat gpurun_out/unfused_stats.txt; } > profiles/${R}_bench_c3_unfused_kernel_stats.txt; fi
for k in fetch write sq lds; do
  { hdr "rocprofv3 --pmc pass '$k' of python3 bench.py --workload c3 --steps 4 --warmup 2 (counters listed in tools/profile_round.sh; one pass per run); SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles; sums over all SIMDs / XCDs; FETCH_SIZE / WRITE_SIZE in KiB"; cat gpurun_out/${T}_$k.txt; } > profiles/${R}_c3_pmc_$k.txt
  { hdr "the same pass for --workload c3w"; cat gpurun_out/${T}_c3w_$k.txt; } > profiles/${R}_c3w_pmc_$k.txt
done
python3 tools/make_traffic_json.py c3 gpurun_out/pmc_${T}_fetch gpurun_out/pmc_${T}_write gpurun_out/pmc_${T}_sq gpurun_out/pmc_${T}_lds > profiles/${R}_traffic_c3.json
python3 tools/make_traffic_json.py c3w gpurun_out/pmc_${T}_c3w_fetch gpurun_out/pmc_${T}_c3w_write gpurun_out/pmc_${T}_c3w_sq gpurun_out/pmc_${T}_c3w_lds > profiles/${R}_traffic_c3w.json
ls profiles | grep "^${R}_"
