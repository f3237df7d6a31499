"""Randomised parity sweep against the oracle (GPU box): the cases of tests/fuzzlib.py: case_small (all input types, sizes around the
path switches, random blockings and sampling rates, forced digit widths and alternative paths).  tests/test_gpu_fuzz.py runs a
fixed-seed slice of it.  python tools/fuzz.py [iterations] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzzlib
from bwtb3m_amd import engine
from oracle import oracle as O
O.build()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); bad = 0
for it in range(iters):
    ok, desc = fuzzlib.case_small(rng, O, engine)
    if not ok:
        bad += 1; print("MISMATCH it=%d %s" % (it, desc), flush=True)
    if it % 25 == 24: print("%d iterations, %d bad, %.0f s" % (it + 1, bad, time.time() - t0), flush=True)
print("done: %d iterations, %d bad" % (iters, bad))
sys.exit(1 if bad else 0)
