set -e
export TMPDIR=/tmp
R=$PWD
rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/r02_stats_v2 -o stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-check --e2e-steps 0 > gpurun_out/r02_stats_v2.log 2>&1
echo stats done
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r02_pmc_fetch -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check --e2e-steps 0 > gpurun_out/r02_pmc_fetch.log 2>&1
echo fetch done
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r02_pmc_write -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check --e2e-steps 0 > gpurun_out/r02_pmc_write.log 2>&1
echo write done
find gpurun_out/r02_stats_v2 gpurun_out/r02_pmc_fetch gpurun_out/r02_pmc_write -name "*.csv" | head -20
find gpurun_out/r02_stats_v2 -name "*agent_info*" -delete; find gpurun_out/r02_stats_v2 -name "*kernel_trace.csv" -delete
