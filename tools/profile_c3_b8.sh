# the merge path at C3 (8 leaf blocks, 3 merge levels: the path north_star names): bench line + rocprofv3 kernel stats; TAG=r04 bash tools/profile_c3_b8.sh
set -e
TAG=${TAG:-r04}
export TMPDIR=/tmp
R=$PWD
ARGS="--blocks 8 --no-cpu-baseline --e2e-steps 0 --no-bwt-only --no-genome-like"
python3 bench.py --steps 5 --warmup 2 $ARGS > gpurun_out/${TAG}_c3_b8_bench.json 2> gpurun_out/${TAG}_c3_b8_bench.err
echo bench done
rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/${TAG}_b8_stats -o stats -- python3 bench.py --steps 5 --warmup 2 --no-check $ARGS > gpurun_out/${TAG}_b8_stats.log 2>&1
find gpurun_out/${TAG}_b8_stats -name "*agent_info*" -delete; find gpurun_out/${TAG}_b8_stats -name "*kernel_trace.csv" -delete
echo stats done
