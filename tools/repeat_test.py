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

# ---- the same text as suffix-range shards (what each GPU of an N-GPU run does; here one after the other on one GPU) ----
nsh = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if nsh > 1:
    plan.close()
    plans, slices, t_run = [], [], []
    for k in range(nsh):
        p = engine.Plan(nsym, "pacterm", shard=k, nshards=nsh, isasamplingrate=1 << 16)
        p.run(d_text); torch.cuda.synchronize()
        t0 = time.perf_counter(); sl = p.run(d_text).clone(); torch.cuda.synchronize(); t_run.append(time.perf_counter() - t0)
        plans.append(p); slices.append(sl)
    pend = [p.tie_pending() for p in plans]
    print("shards %d: local sort max %.1f ms (rounds %d), pending ties %s" % (nsh, max(t_run) * 1e3, max(p.result()["tie_rounds"] for p in plans), pend), flush=True)
    t_fin = []
    if sum(pend):
        exported = [p.tie_export("cuda") for p in plans]
        sa_all = torch.cat([e[0] for e in exported]); pairs_all = torch.cat([e[1] for e in exported])
        for k, p in enumerate(plans):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            engine.finish_shard_ties(p, slices[k], lambda sa: sa_all.clone(), lambda pr, k=k: (pairs_all, sum(pend[:k])))
            torch.cuda.synchronize(); t_fin.append(time.perf_counter() - t0)
        print("collective finish per shard max %.1f ms (rounds in all %d)" % (max(t_fin) * 1e3, max(p.result()["tie_rounds"] for p in plans)), flush=True)
    print("=> per GPU %.1f ms (local sort + finish; the all-gather of %d suffix array entries is not in it)" % ((max(t_run) + (max(t_fin) if t_fin else 0)) * 1e3, nsym), flush=True)
    whole = torch.cat(slices)
    chk = engine.Plan(nsym, "pacterm", isasamplingrate=1 << 16)
    ref = chk.run(d_text)
    print("equals the single-block BWT:", bool(torch.equal(whole, ref)))
