# SQ counters of the three main kernels of ONE C3 step on uniform text (no genome-like leg, no extra legs): four `rocprofv3 --pmc` passes,
# each its own run (gpurun refuses --pmc together with other trace domains); summary -> gpurun_out/${TAG}_sq_counters.json
#   TAG=r04 bash tools/pmc_sq.sh        (copy the JSON into profiles/ afterwards)
export TMPDIR=/tmp
TAG=${TAG:-r04}
R=$PWD
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-check --e2e-steps 0 --no-bwt-only --no-genome-like"
PROG=${PROG:-bench.py}          # (PROG="tools/genome_rounds.py" ARGS="3e9 0 1": the kernels of the list rounds on the genome-like text)
ARGS=${PROGARGS:-$ARGS}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --output-format csv --kernel-trace --pmc $set -d $R/gpurun_out/${TAG}_sq_p$i -o pmc -- python3 $PROG $ARGS > gpurun_out/${TAG}_sq_p$i.log 2>&1 || echo "FAILED pass $i: $set"
  echo "pass $i done"
done
python3 tools/make_sq_json.py $TAG
