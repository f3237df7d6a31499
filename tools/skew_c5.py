"""A skewed text at C5 size on one GPU (sequential suffix-range shards with borders inside level-0 digits: the fine two-pass level 0 + the window test, 36-bit positions):
30 Gbase at 85 % A by default, every LF chain verified.  python tools/skew_c5.py [nbases] [P(A)]"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from bwtb3m_amd import engine
nb = int(float(sys.argv[1])) if len(sys.argv) > 1 else 30_000_000_000
pa = float(sys.argv[2]) if len(sys.argv) > 2 else 0.85
nb -= nb % 4
g = torch.Generator(device="cuda"); g.manual_seed(11)
rest = (1.0 - pa) / 3.0
parts = []
t0 = time.time()
for o in range(0, nb, 1 << 28):
    m = min(1 << 28, nb - o)
    u = torch.rand(m, device="cuda", generator=g)
    b = (u >= pa).to(torch.uint8) + (u >= pa + rest).to(torch.uint8) + (u >= pa + 2 * rest).to(torch.uint8)
    bb = b.view(-1, 4)
    parts.append(((bb[:, 0] << 6) | (bb[:, 1] << 4) | (bb[:, 2] << 2) | bb[:, 3]).to(torch.uint8))
    del u, b, bb
d_text = torch.cat(parts + [torch.zeros(66, dtype=torch.uint8, device="cuda")]); del parts
n = nb + 1
print("text of %d symbols at %.0f %% A generated in %.1f s" % (n, pa * 100, time.time() - t0), flush=True)
plan = engine.Plan(n, "pacterm", isasamplingrate=1 << 16)
out = torch.empty(n, dtype=torch.uint8, device="cuda")
t0 = time.perf_counter(); plan.run(d_text, out); torch.cuda.synchronize(); dt = time.perf_counter() - t0
r = plan.result()
print("sorted in %.2f s (%.2f Gbase/s), blocks %d, tie rounds %d, listed %d" % (dt, nb / dt / 1e9, r["num_blocks"], r["tie_rounds"], r["tie_elements"]), flush=True)
print("   ", {k["name"]: round(k["ms"], 1) for k in plan.kernel_times() if k["ms"] > 1.0}, flush=True)
t0 = time.perf_counter(); m0, m1, _ = plan.lf_walk(out, sarate=0); torch.cuda.synchronize()
print("LF walk over all chains: %d symbol / %d chain mismatches (%.1f s); BWT[p0] = %d" % (m0, m1, time.perf_counter() - t0, int(out[r["p0"]])), flush=True)
