"""profiles/rNN_pmc_traffic.json from two rocprofv3 --pmc passes (gpurun_out/<tag>_pmc_fetch, gpurun_out/<tag>_pmc_write):
python tools/make_pmc_json.py r02 "kernel set description"."""
import csv, glob, collections, json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
desc = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    fs = sorted(glob.glob("/root/repo/gpurun_out/%s_pmc_%s/**/*counter_collection.csv" % (tag, kind), recursive=True))
    for r in csv.DictReader(open(fs[-1])):
        if r["Counter_Name"] != ctr: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "b3m::" not in k: continue
        a = acc[k][ctr]; a[0] += float(r["Counter_Value"]); a[1] += 1
out = {"workload_n": 3000000001,
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of `python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check --e2e-steps 0` (C3, one block), " + desc,
       "units": "FETCH_SIZE/WRITE_SIZE are KiB; read bytes = 2 x FETCH_SIZE x 1024 (gfx950 reports half of a coalesced stream, MI355X_MICROARCH.md HBM section; round 1 calibrated it on a pass that reads exactly 24.0 GB), write bytes = WRITE_SIZE x 1024",
       "kernels": {}}
tot = 0.0
for k, v in acc.items():
    f, w = v.get("FETCH_SIZE", [0, 0]), v.get("WRITE_SIZE", [0, 0])
    n = max(f[1], w[1], 1)
    rd, wr = 2 * f[0] * 1024 / n, w[0] * 1024 / n
    tot += (rd + wr) * n
    out["kernels"][k] = {"launches": n, "fetch_size_kib_per_launch": f[0] / n, "write_size_kib_per_launch": w[0] / n,
                         "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
out["total_hbm_bytes_per_step"] = tot
json.dump(out, open("/root/repo/profiles/%s_pmc_traffic.json" % tag, "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:8]:
    print("%-40s x%d  read %.2f GB  write %.2f GB" % (k, v["launches"], v["hbm_read_bytes_per_launch"] / 1e9, v["hbm_write_bytes_per_launch"] / 1e9))
print("total %.1f GB per step" % (tot / 1e9))
