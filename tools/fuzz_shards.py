"""Randomised sweep of the suffix-range shards against the oracle (tests/fuzzlib.py: case_shards): the slices of N shards must tile the
BWT, their samples must be the oracle's inverse suffix array.  python tools/fuzz_shards.py [iterations] [seed]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzzlib
from bwtb3m_amd import engine
from oracle import oracle as O
O.build()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = 0; t0 = time.time()
for it in range(iters):
    ok, desc = fuzzlib.case_shards(rng, O, engine, torch)
    if not ok: bad += 1; print("MISMATCH it=%d %s" % (it, desc), flush=True)
    if it % 100 == 99: print("%d iterations, %d bad, %.0f s" % (it + 1, bad, time.time() - t0), flush=True)      # (a run that writes nothing for minutes is taken to be hung)
print("done: %d iterations, %d bad, %.0f s" % (iters, bad, time.time() - t0))
sys.exit(1 if bad else 0)
