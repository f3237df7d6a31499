"""Byte alphabets at sizes the oracle cannot sort (tests/fuzzlib.py: case_bytes_large): the one-block BWT against blocks of a merge tree.
python tools/fuzz_bytes_large.py [iterations] [seed]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzzlib
from bwtb3m_amd import engine
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 15
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
bad = 0; t0 = time.time()
for it in range(iters):
    ok, desc = fuzzlib.case_bytes_large(rng, engine, torch)
    print("%s it=%d %s  %.0f s" % ("ok " if ok else "BAD", it, desc, time.time() - t0), flush=True)
    bad += 0 if ok else 1
print("done: %d iterations, %d bad" % (iters, bad))
sys.exit(1 if bad else 0)
