"""Base-composition skew (uniform, human-like 59 % A+T, 65 %, 80 %) at 1 Gbase: time, listed suffixes, LF check.
python tools/skew_test.py [n] [which: comma list of 0..3]"""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from bwtb3m_amd import engine
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
which = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3]
def run(name, p):
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    u = torch.rand(N, device="cuda", generator=g)
    cp = np.cumsum(p)
    b = (u >= cp[0]).to(torch.uint8) + (u >= cp[1]).to(torch.uint8) + (u >= cp[2]).to(torch.uint8)
    del u
    b = b.view(-1, 4) if N % 4 == 0 else None
    pac = ((b[:, 0] << 6) | (b[:, 1] << 4) | (b[:, 2] << 2) | b[:, 3]).contiguous()
    d_text = torch.cat([pac, torch.zeros(2, dtype=torch.uint8, device="cuda")])   # l/4 + 2 bytes, last byte = l % 4 = 0
    del b, pac
    n = N + 1
    plan = engine.Plan(n, "pacterm", isasamplingrate=1 << 16)
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        plan.run(d_text, out); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    r = plan.result()
    m0, m1, _ = plan.lf_walk(out, sarate=0)
    print("%-22s n=%d  %.1f ms  rounds %d listed %d  LF %d %d" % (name, n, dt * 1e3, r["tie_rounds"], r["tie_elements"], m0, m1), flush=True)
    print("   ", {k["name"]: round(k["ms"], 1) for k in plan.kernel_times() if k["ms"] > 0.3}, flush=True)
    plan.close()
cases = [("uniform", [0.25, 0.25, 0.25, 0.25]), ("AT 59 % (human)", [0.295, 0.205, 0.205, 0.295]), ("AT 65 %", [0.325, 0.175, 0.175, 0.325]),
         ("AT 80 % (P. falciparum)", [0.40, 0.10, 0.10, 0.40])]
for i in which:
    run(*cases[i])
