"""Byte alphabets at sizes the oracle cannot sort (5 M - 80 M symbols): the one-block BWT against blocks of a merge tree (count-row rank
dictionary, byte gap walk), random alphabets (2 .. 256 symbols, skewed) and planted repeats.  No ground truth at these sizes: the two
forms must agree, and the symbol histogram of the BWT must be the text's.  python tools/fuzz_bytes_large.py [iterations] [seed]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from bwtb3m_amd import engine
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 15
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
bad = 0; t0 = time.time()
for it in range(iters):
    n = int(rng.integers(5_000_000, 80_000_000)); sig = int([2, 4, 26, 64, 256][int(rng.integers(0, 5))])
    g = torch.Generator(device="cuda"); g.manual_seed(int(rng.integers(1, 1 << 40)))
    t = torch.randint(0, sig, (n,), device="cuda", dtype=torch.int32, generator=g)
    if rng.integers(0, 2): t = (t * t) % sig                      # skewed symbol frequencies
    t = (t + int(rng.integers(0, 256 - sig + 1))).to(torch.uint8)
    for _ in range(int(rng.integers(0, 4))):
        ln = int(rng.integers(100, 200_000)); dst = int(rng.integers(0, n - ln))
        if rng.integers(0, 2): src = int(rng.integers(0, n - ln)); t[dst:dst + ln] = t[src:src + ln].clone()
        else: u = int(rng.integers(1, 100)); t[dst:dst + ln] = t[dst:dst + u].repeat((ln + u - 1) // u)[:ln]
    nblk = int(rng.integers(2, 7)); ok = True; msg = ""
    try:
        d_text = torch.cat([t, torch.zeros(64, dtype=torch.uint8, device="cuda")])
        p1 = engine.Plan(n, "bytestream"); one = p1.run(d_text).clone(); torch.cuda.synchronize(); p1.close()
        pb = engine.Plan(n, "bytestream", blocksize=(n + nblk - 1) // nblk); out = pb.run(d_text); torch.cuda.synchronize()
        ok &= bool(torch.equal(out, one)) and bool(torch.equal(torch.bincount(one.to(torch.int64), minlength=256), torch.bincount(t.to(torch.int64), minlength=256)))
        msg = "blocks %d" % pb.result()["num_blocks"]; pb.close()
    except Exception as ex:   # noqa: BLE001
        ok = False; msg = "EXC " + repr(ex)[:300]
    print("%s it=%d n=%d sigma=%d %s  %.0f s" % ("ok " if ok else "BAD", it, n, sig, msg, time.time() - t0), flush=True)
    bad += 0 if ok else 1
    del t, d_text; torch.cuda.empty_cache()
print("done: %d iterations, %d bad" % (iters, bad))
sys.exit(1 if bad else 0)
