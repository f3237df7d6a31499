"""One-off stress runs at scale (LF-walk check, timing): degenerate inputs through the skew paths."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from bwtb3m_amd import engine
from oracle import oracle as O

def run(name, b):
    buf = O.pac_encode(b); n = len(b) + 1
    d_text = torch.from_numpy(np.asarray(buf)).cuda()
    plan = engine.Plan(n, "pacterm", isasamplingrate=1 << 16)
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan.run(d_text, out); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    r = plan.result()
    m0, m1, _ = plan.lf_walk(out, sarate=0)
    print("%-28s n=%d  %.1f ms  rounds %d listed %d  LF mismatches %d %d" % (name, n, dt * 1e3, r["tie_rounds"], r["tie_elements"], m0, m1), flush=True)
    print("   ", {k["name"]: round(k["ms"], 1) for k in plan.kernel_times() if k["ms"] > 0.5}, flush=True)
    plan.close()

rng = np.random.default_rng(5)
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
run("all A", np.zeros(N, dtype=np.uint8))
run("two symbols", rng.integers(0, 2, N, dtype=np.uint8))
b = rng.integers(0, 4, N, dtype=np.uint8); b[N // 4: N // 2] = np.resize(np.array([0, 1, 2, 3, 1], dtype=np.uint8), N // 4)
run("quarter period-5", b)
p = np.array([0.7, 0.1, 0.1, 0.1]); run("skewed 70/10/10/10", rng.choice(4, N, p=p).astype(np.uint8))
