"""Randomised parity sweep against the oracle (GPU box): sizes around the path switches (2^18 = MSD threshold,
block edges not aligned to words / chunks), all input types, random blockings, low-entropy and repeat-rich texts.
python tools/fuzz.py [iterations] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from bwtb3m_amd import engine
from oracle import oracle as O
O.build()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); bad = 0
for it in range(iters):
    itype = ["pac", "pacterm", "bytestream"][int(rng.integers(0, 3))]
    pick = rng.integers(0, 5)
    nb = int([rng.integers(1, 2000), rng.integers(260000, 270000), rng.integers(250000, 700000), rng.integers(1, 100000), rng.integers(520000, 1100000)][pick])
    kind = int(rng.integers(0, 4))
    if itype == "bytestream":
        sig = int([256, 4, 2, 26][kind])
        t = rng.integers(0, sig, nb, dtype=np.uint8)
        if kind == 3 and nb > 100: k = int(rng.integers(1, nb // 2)); t[nb - k:] = t[:k]
        buf, s = t, t
    else:
        b = rng.integers(0, [4, 4, 2, 4][kind], nb, dtype=np.uint8)
        if kind == 1 and nb > 100: k = int(rng.integers(1, nb // 2)); p = int(rng.integers(0, nb - k)); b[p:p + k] = 0
        if kind == 3 and nb > 100: k = int(rng.integers(1, nb // 2)); b[nb - k:] = b[:k]
        buf = O.pac_encode(b); s = O.symbols_from_file_bytes(buf, itype)
    n = len(s)
    bs = 0
    if rng.integers(0, 2) and n > 20:              # blocks of a merge tree (2-bit and byte alphabets)
        bs = int(rng.integers(max(8, n // 9), n))
    # level-1 width forced on half of the cases: small inputs through the kernels of the large geometries (B3M_MSD_D1, see k_msd.hip)
    d1 = int([0, 0, 0, 2, 4, 5, 6, 8][int(rng.integers(0, 8))])
    if d1: os.environ["B3M_MSD_D1"] = str(d1)
    else: os.environ.pop("B3M_MSD_D1", None)
    isar = int([0, 64, 4096][int(rng.integers(0, 3))])
    try:
        out, res, plan = engine.bwt_device(buf, itype, blocksize=bs, isasamplingrate=isar)
        ref = O.bwt(s)
        ok = np.array_equal(out, ref)
        if ok and isar:
            sa = O.sa_circular(s); smp = plan.samples()
            ok = np.array_equal(sa[smp[:, 0].astype(np.int64)], smp[:, 1].astype(sa.dtype))
        plan.close()
    except Exception as ex:   # noqa: BLE001
        ok = False; print("EXC", repr(ex)[:200])
    if not ok:
        bad += 1
        print("MISMATCH it=%d type=%s nb=%d kind=%d bs=%d isar=%d d1=%d" % (it, itype, nb, kind, bs, isar, d1), flush=True)
    if it % 25 == 24: print("%d iterations, %d bad, %.0f s" % (it + 1, bad, time.time() - t0), flush=True)
print("done: %d iterations, %d bad" % (iters, bad))
sys.exit(1 if bad else 0)
