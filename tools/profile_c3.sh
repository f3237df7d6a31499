# the rocprofv3 runs behind profiles/<TAG>_*: kernel stats (--kernel-trace --stats), then FETCH_SIZE and WRITE_SIZE in PMC passes of their own
# (gpurun refuses --pmc together with other trace domains); TAG=r03 bash tools/profile_c3.sh
set -e
TAG=${TAG:-r03}
export TMPDIR=/tmp
R=$PWD
ARGS="--no-cpu-baseline --no-check --e2e-steps 0 --no-bwt-only --no-genome-like"
rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats -o stats -- python3 bench.py --steps 10 --warmup 3 $ARGS > gpurun_out/${TAG}_stats.log 2>&1
echo stats done
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_pmc_fetch -o pmc -- python3 bench.py --steps 1 --warmup 0 $ARGS > gpurun_out/${TAG}_pmc_fetch.log 2>&1
echo fetch done
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_pmc_write -o pmc -- python3 bench.py --steps 1 --warmup 0 $ARGS > gpurun_out/${TAG}_pmc_write.log 2>&1
echo write done
find gpurun_out/${TAG}_stats -name "*agent_info*" -delete; find gpurun_out/${TAG}_stats -name "*kernel_trace.csv" -delete
find gpurun_out/${TAG}_stats gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write -name "*.csv" | head -20
