export TMPDIR=/tmp
R=$PWD
for set in "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --output-format csv --kernel-trace --pmc $set -d $R/gpurun_out/pmc_sq_$tag -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check --e2e-steps 0 --no-bwt-only > gpurun_out/pmc_sq_$tag.log 2>&1 || echo "FAILED $set"
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_sq_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "b3m::" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in ("b3m::k_finish_v3<3>", "b3m::k_l0_pass<false, 65280, 4, 2048, true>", "b3m::k_l1_gather<1024, 4>"):
    print(k); 
    for c, v in sorted(acc[k].items()): print("   %-28s %.4g" % (c, v))
PY
