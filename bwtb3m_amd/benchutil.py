"""Shared by bench.py (N=1) and distributed.bench_main (N>1): the `roofline` object of the bench line
for the dominant kernel family, from the engine's own HIP-event timeline (b3m_plan_kernel_times)."""
import json
import os

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s

# rocprof kernel name of each family's main kernel (profiles/*kernel_stats.csv)
KERNEL_OF = {"msd_scatter": "b3m::k_l1_gather<1024, 4>", "msd_scatter_text": "b3m::k_l0_pass<false, 65280, 4, 2048, true, 0, false>",
             "msd_hist": "b3m::k_l0_pass<true, 65280, 4, 2048, false, 0, false>", "bucket_finish": "b3m::k_finish_v3<3, 512, false>", "ties": "b3m::k_tie_fix",
             "radix_scatter": "b3m::k_scatter<unsigned int, false, 512>", "gap_walk": "b3m::k_gap_walk4"}


def pmc_kernel_name(name, kernels):
    """the entry of a PMC file's `kernels` for `name`: the exact instantiation, else the instantiation of the same kernel template that
    moved the most bytes (a template parameter added later must not turn `roofline.traffic` into null: round 4 did)"""
    if name in kernels:
        return name
    base = name.split("<")[0]
    same = [k for k in kernels if k.split("<")[0] == base]
    return max(same, key=lambda k: kernels[k].get("hbm_bytes_per_launch", 0)) if same else None


def add_kernel_times(fam, kernel_times):
    for k in kernel_times:
        f = fam.setdefault(k["name"], {"ms": 0.0, "launches": 0, "alg_bytes": 0})
        f["ms"] += k["ms"]; f["launches"] += k["launches"]; f["alg_bytes"] += k["alg_bytes"]


def dominant_roofline(fam, steps, dt, nbases, levels, pmc_n, root, peak_gpus=1, shards=1):
    """fam: accumulated over `steps` runs on this rank; dt: wall seconds of those steps; pmc_n: symbol
    count for which the committed PMC traffic may be quoted (0 = none); shards: suffix-range shards of the run (their level 0 is the
    one-pass instantiation of k_l0_pass: 512-thread workgroups from three shards on)."""
    dname, d = max(fam.items(), key=lambda kv: kv[1]["ms"])
    if shards > 1 and dname == "msd_scatter_text":
        KERNEL_OF_ = dict(KERNEL_OF, msd_scatter_text="b3m::k_l0_pass<false, 65280, 4, 2048, true, %d, false>" % (2 if shards >= 3 else 1))
    else:
        KERNEL_OF_ = KERNEL_OF
    achieved = d["alg_bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0
    # HBM traffic of the dominant kernel from the committed PMC pass (separate rocprofv3 --pmc runs,
    # gfx950 FETCH_SIZE correction applied: see the file); only valid for the workload it was taken on
    traffic = None
    try:
        import glob
        files = sorted(glob.glob(os.path.join(root, "profiles", "r0*_pmc_traffic.json")))      # the latest round's PMC passes
        tj = json.load(open(files[-1])) if files else {}
        kn = pmc_kernel_name(KERNEL_OF_.get(dname, dname), tj.get("kernels", {}))
        if pmc_n and tj.get("workload_n") == pmc_n and kn is not None:
            traffic = int(tj["kernels"][kn]["hbm_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        traffic = None
    alg_job = nbases * (9.5 + 41.0 * levels)      # alg_bytes(n,B)
    return {
        "bound": "hbm", "kernel": KERNEL_OF_.get(dname, dname), "family": dname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
        "avg_launch_ms": round(d["ms"] / max(1, d["launches"]), 4), "launches_per_step": d["launches"] // max(1, steps),
        "alg_bytes_per_launch": d["alg_bytes"] // max(1, d["launches"]),
        "job_alg_bytes": alg_job, "job_frac": round(alg_job / (dt / steps) / 1e9 / (HBM_PEAK_GBS * peak_gpus), 5),
        "families_ms_per_step": {k: round(v["ms"] / steps, 3) for k, v in fam.items() if v["ms"] > 0},
    }


def text_histogram(d_text, n, itype):
    """symbol histogram of a packed input tensor, computed with torch alone (independent of the engine):
    256-entry int64 list in the output alphabet (pacterm: base + 1, one terminator 0)"""
    import torch
    if itype == "bytestream":
        h = torch.zeros(256, dtype=torch.int64, device=d_text.device)
        for o in range(0, n, 1 << 28):
            h += torch.bincount(d_text[o:min(n, o + (1 << 28))].to(torch.int64), minlength=256)
        return h.cpu().tolist()
    nb = n - 1 if itype == "pacterm" else n
    full = nb // 4
    hb = torch.zeros(256, dtype=torch.int64, device=d_text.device)
    for o in range(0, full, 1 << 28):
        hb += torch.bincount(d_text[o:min(full, o + (1 << 28))].to(torch.int64), minlength=256)
    hb = hb.cpu().tolist()
    base = [0, 0, 0, 0]
    for byte, c in enumerate(hb):
        if c:
            for k in range(4):
                base[(byte >> (2 * k)) & 3] += c
    if nb % 4:
        last = int(d_text[full])
        for k in range(nb % 4):
            base[(last >> (6 - 2 * k)) & 3] += 1
    out = [0] * 256
    if itype == "pacterm":
        out[0] = 1
        for c in range(4):
            out[c + 1] = base[c]
    else:
        for c in range(4):
            out[c] = base[c]
    return out
