"""Timing of the overflow fallback on a repeat-rich synthetic genome (not a parity test: LF-walk check only)."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from bwtb3m_amd import engine, synth
from oracle import oracle as O

def make(n, seed):
    rng = np.random.default_rng(seed)
    b = rng.integers(0, 4, n, dtype=np.uint8)
    def mutate(seg, rate):
        s = seg.copy(); k = int(len(s) * rate)
        if k: idx = rng.integers(0, len(s), k); s[idx] = (s[idx] + rng.integers(1, 4, k)) & 3
        return s
    alu = rng.integers(0, 4, 300, dtype=np.uint8)
    for _ in range(n // 3000):                       # ~10 % of the text: 300-bp element, 10 % divergence
        p = int(rng.integers(0, n - 300)); b[p:p + 300] = mutate(alu, 0.10)
    l1 = rng.integers(0, 4, 6000, dtype=np.uint8)
    for _ in range(n // 120000):                     # ~5 %: 6-kb element, 3 % divergence
        p = int(rng.integers(0, n - 6000)); b[p:p + 6000] = mutate(l1, 0.03)
    sd = rng.integers(0, 4, 200000, dtype=np.uint8)
    for _ in range(6):                               # segmental duplications: 200 kb, 0.1 % divergence
        p = int(rng.integers(0, n - 200000)); b[p:p + 200000] = mutate(sd, 0.001)
    for _ in range(n // 100000):                     # microsatellites
        u = int(rng.integers(1, 7)); c = int(rng.integers(20, 200)); p = int(rng.integers(0, n - u * c))
        b[p:p + u * c] = np.resize(rng.integers(0, 4, u, dtype=np.uint8), u * c)
    sat = rng.integers(0, 4, 171, dtype=np.uint8)    # satellite array: 171-bp unit x 3000, 1 % divergence
    p = int(rng.integers(0, n - 171 * 3000)); b[p:p + 171 * 3000] = mutate(np.resize(sat, 171 * 3000), 0.01)
    return b

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
blocksize = int(float(sys.argv[2])) if len(sys.argv) > 2 else 0      # > 0: blocks of a merge tree (B3M_NO_TREE_DOUBLING=1 for the rounds without doubling)
b = make(n, 7)
buf = O.pac_encode(b) if hasattr(O, "pac_encode") else None
if buf is None:
    raise SystemExit("no pac encoder")
d_text = torch.from_numpy(np.asarray(buf)).cuda()
nsym = n + 1
plan = engine.Plan(nsym, "pacterm", isasamplingrate=1 << 16, blocksize=blocksize)
out = torch.empty(nsym, dtype=torch.uint8, device="cuda")
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan.run(d_text, out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    r = plan.result()
    print("run %d: %.1f ms  blocks %d tie_rounds %d tie_elements %d overhang %d slow %d" % (it, dt * 1e3, r["num_blocks"], r["tie_rounds"], r["tie_elements"], r["tie_overhang"], r["tie_slow"]), flush=True)
    print({k["name"]: round(k["ms"], 2) for k in plan.kernel_times() if k["ms"] > 0.05}, flush=True)
m0, m1, _ = plan.lf_walk(out, sarate=0)
print("LF check mismatches:", m0, m1)
