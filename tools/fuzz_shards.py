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
    parts = []; off = 0; ok = True
    try:
        for k in range(ns):
            plan = engine.Plan(n, itype, shard=k, nshards=ns)
            sl = plan.run(d_text); torch.cuda.synchronize()
            r = plan.result()
            ok &= r["slice_offset"] == off
            parts.append(sl.cpu().numpy()); off += sl.numel(); plan.close()
        ok &= off == n and np.array_equal(np.concatenate(parts), ref)
    except Exception as ex:   # noqa: BLE001
        ok = False; print("EXC", repr(ex)[:200])
    if not ok: bad += 1; print("MISMATCH it=%d type=%s nb=%d kind=%d shards=%d" % (it, itype, nb, kind, ns), flush=True)
print("done: %d iterations, %d bad, %.0f s" % (iters, bad, time.time() - t0))
sys.exit(1 if bad else 0)
