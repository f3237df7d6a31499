"""Randomised sweep of the suffix-range shards against the oracle: the slices of N shards must tile the BWT."""
import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from bwtb3m_amd import engine
from oracle import oracle as O
O.build()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = 0; t0 = time.time()
for it in range(iters):
    itype = ["pac", "pacterm"][int(rng.integers(0, 2))]
    nb = int(rng.integers(262200, 900000))
    kind = int(rng.integers(0, 3))
    b = rng.integers(0, [4, 4, 2][kind], nb, dtype=np.uint8)
    if kind == 1: k = int(rng.integers(1, nb // 3)); p = int(rng.integers(0, nb - k)); b[p:p + k] = b[:k]
    buf = O.pac_encode(b); s = O.symbols_from_file_bytes(buf, itype); n = len(s)
    ref = O.bwt(s)
    ns = int(rng.integers(2, 9))
    d_text = torch.from_numpy(np.asarray(buf)).cuda()
    # half of the cases with forced digit widths (test switches of k_msd.hip): the kernels of the large geometries, and with few
    # level-0 bits the shards' fine level 0
    import os
    geo = int(rng.integers(0, 4))
    for k_, v_ in (("B3M_MSD_D0", [None, None, "2", "3"][geo]), ("B3M_MSD_D1", [None, "4", "3", "4"][geo])):
        if v_ is None: os.environ.pop(k_, None)
        else: os.environ[k_] = v_
    ok = True
    try:
        plans, slices = [], []
        for k in range(ns):
            plan = engine.Plan(n, itype, shard=k, nshards=ns)
            slices.append(plan.run(d_text).clone()); plans.append(plan)
        torch.cuda.synchronize()
        pend = [p.tie_pending() for p in plans]
        if sum(pend):                              # long repeats: the collective finish over all shards' suffix arrays
            for k, p in enumerate(plans):
                if not p.has_sa():
                    sl = torch.empty(p.slice_capacity, dtype=torch.uint8, device="cuda")
                    engine.rerun_for_ties(p, d_text, sl)
                    slices[k] = sl[: p.result()["slice_len"]].clone()
            pend = [p.tie_pending() for p in plans]
            exported = [p.tie_export("cuda") for p in plans]
            sa_all = torch.cat([e[0] for e in exported]); pairs_all = torch.cat([e[1] for e in exported])
            for k, p in enumerate(plans):
                engine.finish_shard_ties(p, slices[k], lambda sa: sa_all.clone(), lambda pr, k=k: (pairs_all, sum(pend[:k])))
        off = 0
        for p, sl in zip(plans, slices):
            r = p.result()
            ok &= r["slice_offset"] == off and r["slice_len"] == sl.numel()
            off += sl.numel()
        ok &= off == n and np.array_equal(torch.cat(slices).cpu().numpy(), ref)
        for p in plans: p.close()
    except Exception as ex:   # noqa: BLE001
        ok = False; print("EXC", repr(ex)[:200])
    if not ok: bad += 1; print("MISMATCH it=%d type=%s nb=%d kind=%d shards=%d geo=%d" % (it, itype, nb, kind, ns, geo), flush=True)
print("done: %d iterations, %d bad, %.0f s" % (iters, bad, time.time() - t0))
sys.exit(1 if bad else 0)
