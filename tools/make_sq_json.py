"""gpurun_out/<tag>_sq_counters.json from the rocprofv3 --pmc passes of tools/pmc_sq.sh (gpurun_out/<tag>_sq_p*): per kernel of the record sort,
the SQ counters summed over its launches of the ONE timed step, its duration under the counters, and the figures derived from them."""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
WANT = ("k_finish_v3", "k_l1_gather", "k_l0_pass", "k_finish_fast", "k_tie_fix_list", "k_gap_walk4", "k_u32_pass", "k_u32_finish", "k_interleave", "k_tie_tile", "k_msd_scatter", "k_msd_count", "k_msd_repid", "k_tie_keyvals", "k_isa_slots", "k_bucket_finish")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
meta = {}
for f in sorted(glob.glob("gpurun_out/%s_sq_p*/**/*counter_collection.csv" % tag, recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "b3m::" not in k or not any(w in k for w in WANT):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        d = (r["Dispatch_Id"])
        m = meta.setdefault(k, {"launches": 0, "ns_under_counters": 0.0, "workgroup": int(r["Workgroup_Size"]), "grid": int(r["Grid_Size"]),
                                "lds_bytes": int(r["LDS_Block_Size"]), "vgprs": int(r["VGPR_Count"]), "sgprs": int(r["SGPR_Count"]), "_f": f})
        if m["_f"] == f and d not in seen:
            seen.add(d)
            m["launches"] += 1
            m["ns_under_counters"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {"workload": "C3 (n = 3,000,000,001, uniform pacterm), one block, ONE step of `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check --e2e-steps 0 "
                   "--no-bwt-only --no-genome-like` per pass (tools/pmc_sq.sh)",
       "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over all waves (MI355X_MICROARCH.md); SQ_BUSY_CYCLES per SE; "
                "WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES",
       "kernels": {}}
for k, c in acc.items():
    m = dict(meta[k])
    m.pop("_f")
    d = {"counters": dict(sorted(c.items()))}
    d.update(m)
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    der = {}
    if wc and gui:
        # resident waves per CU, averaged over the kernel: wave-cycles (x4: quad-cycles) / (GPU-active cycles x 256 CUs); GRBM_GUI_ACTIVE is summed over the 8 XCDs
        der["avg_resident_waves_per_cu"] = round(4.0 * wc / (gui / 8.0 * 256.0), 2)
    for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA"):
        if wc and name in c:
            der[name.lower() + "_over_wave_cycles"] = round(c[name] / wc, 4)
    if c.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in c:
        der["lds_bank_conflict_over_lds_active"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
    if c.get("SQ_WAVES"):
        for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if name in c:
                der[name.lower() + "_per_wave"] = round(c[name] / c["SQ_WAVES"], 1)
    d["derived"] = der
    out["kernels"][k] = d
json.dump(out, open("gpurun_out/%s_sq_counters.json" % tag, "w"), indent=1)
for k, d in out["kernels"].items():
    print(k, "x%d" % d["launches"], "%.2f ms" % (d["ns_under_counters"] / 1e6), json.dumps(d["derived"]))
