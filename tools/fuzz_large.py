"""Randomised sweep at sizes the oracle cannot sort (tests/fuzzlib.py: case_large): random composition, planted repeats, one block /
blocks of a merge tree / mem=-chosen blocks / suffix-range shards / host-to-host; every result LF-verified on the device, the forms
compared with each other.  python tools/fuzz_large.py [iterations] [seed] [min bases] [max bases]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzzlib
from bwtb3m_amd import engine
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
lo = int(float(sys.argv[3])) if len(sys.argv) > 3 else 24_000_000
hi = int(float(sys.argv[4])) if len(sys.argv) > 4 else 400_000_000
bad = 0; t0 = time.time()
for it in range(iters):
    ok, desc = fuzzlib.case_large(rng, engine, torch, lo, hi)
    print("%s it=%d %s  %.0f s" % ("ok " if ok else "BAD", it, desc, time.time() - t0), flush=True)
    bad += 0 if ok else 1
print("done: %d iterations, %d bad" % (iters, bad))
sys.exit(1 if bad else 0)
