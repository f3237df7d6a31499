import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fuzzlib
from bwtb3m_amd import engine
alt = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tie = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nb = 307623469
rng = np.random.default_rng(99)
g = torch.Generator(device="cuda"); g.manual_seed(12345)
b = torch.empty(nb, dtype=torch.uint8, device="cuda")
for o in range(0, nb, 1 << 28):
    m_ = min(1 << 28, nb - o)
    b[o:o + m_] = torch.randint(0, 4, (m_,), device="cuda", generator=g, dtype=torch.int64).to(torch.uint8)
b[194290333:194290333 + 96082] = 3
b[1928638:1928638 + 94826] = 2
d_text = fuzzlib.pac_image(torch, b)
n = nb + 1
fuzzlib._set_env(B3M_NO_FAST_FINISH=1 if alt == 2 else None, B3M_NO_TIE_MSD=1 if tie == 5 else None)
plan = engine.Plan(n, "pacterm", isasamplingrate=65536)
one = plan.run(d_text).clone(); torch.cuda.synchronize()
print("LF", plan.lf_walk(one, sarate=0)[:2], plan.result()["tie_rounds"], flush=True)
plan.close()
plan = engine.Plan(n, "pacterm", isasamplingrate=65536)
h_text = torch.zeros(plan.text_bytes, dtype=torch.uint8).pin_memory(); h_text.copy_(d_text[:plan.text_bytes].cpu())
h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
for it in range(2):
    plan.run_host(h_text, h_out)
    diff = (h_out != one.cpu()).nonzero().flatten()
    print("byte run", it, "mismatches", diff.numel(), diff[:5].tolist(), diff[-5:].tolist() if diff.numel() else "", flush=True)
h_pk = torch.empty((n + 3) // 4, dtype=torch.uint8).pin_memory()
for it in range(2):
    plan.run_host(h_text, h_pk, packed=True)
    ref = engine.Plan.pack_pac(one, True).cpu()
    diff = (h_pk != ref).nonzero().flatten()
    print("packed run", it, "mismatches", diff.numel(), diff[:5].tolist(), diff[-5:].tolist() if diff.numel() else "", "p0", plan.result()["p0"], flush=True)
