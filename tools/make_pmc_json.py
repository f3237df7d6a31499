"""profiles/r01_pmc_traffic.json from two rocprofv3 --pmc passes (gpurun_out/pmc_fetch4, gpurun_out/pmc_write4)."""
import csv, glob, collections, json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "4"
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    fs = sorted(glob.glob("/root/repo/gpurun_out/pmc_%s%s/**/*counter_collection.csv" % (kind, tag), recursive=True))
    for r in csv.DictReader(open(fs[-1])):
        if r["Counter_Name"] != ctr: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "b3m::" not in k: continue
        a = acc[k][ctr]; a[0] += float(r["Counter_Value"]); a[1] += 1
out = {"workload_n": 3000000001,
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of `python bench.py --steps 1 --warmup 0 --no-cpu-baseline` (C3, one block), round 1, kernel set v5 (11-bit level 0 from LDS-sorted indices in 65,280-position chunks, 8-bit level 1 in 16 Ki-record chunks, in-LDS finish)",
       "units": "FETCH_SIZE/WRITE_SIZE are KiB; read bytes = 2 x FETCH_SIZE x 1024 (gfx950 reports half of a coalesced stream, MI355X_MICROARCH.md HBM section; calibrated on the level-1 counting pass, which reads exactly 24.0 GB), write bytes = WRITE_SIZE x 1024",
       "kernels": {}}
for k, v in acc.items():
    f, w = v.get("FETCH_SIZE", [0, 0]), v.get("WRITE_SIZE", [0, 0])
    n = max(f[1], w[1], 1)
    rd, wr = 2 * f[0] * 1024 / n, w[0] * 1024 / n
    out["kernels"][k] = {"launches": n, "fetch_size_kib_per_launch": f[0] / n, "write_size_kib_per_launch": w[0] / n,
                         "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
json.dump(out, open("/root/repo/profiles/r01_pmc_traffic.json", "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:8]:
    print("%-40s x%d  read %.2f GB  write %.2f GB" % (k, v["launches"], v["hbm_read_bytes_per_launch"] / 1e9, v["hbm_write_bytes_per_launch"] / 1e9))
