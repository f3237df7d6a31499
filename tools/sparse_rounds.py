"""Long repeats in plans that run suffix-range shards one after the other: rounds and time of the list finish (tie_rounds_sparse) against key extension to the end
(B3M_SPARSE_TIES=0, round 3's form).  python tools/sparse_rounds.py [nbases] [copies] [segment bases]   (host tier forced: B3M_FORCE_SPILL=4)"""
import os, sys, time
os.environ.setdefault("B3M_FORCE_SPILL", "4")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bwtb3m_amd import engine, synth
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000_000
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 40
seglen = int(sys.argv[3]) if len(sys.argv) > 3 else 200_000
itype, d_text, n = synth.config_input_device("c3", size=nb, device="cuda")
d_text = d_text.clone()
sb = seglen // 4
seg = d_text[1_000_000:1_000_000 + sb].clone()
step = (nb // 4 - 4_000_000 - sb) // copies
for c in range(copies):
    at = 3_000_000 + c * step + 17 * c
    d_text[at:at + sb] = seg
res = {}
for cap in ("", "0"):
    if cap: os.environ["B3M_SPARSE_TIES"] = cap
    else: os.environ.pop("B3M_SPARSE_TIES", None)
    plan = engine.Plan(n, itype, isasamplingrate=1 << 14)
    out = plan.run(d_text); torch.cuda.synchronize()
    t0 = time.time(); out = plan.run(d_text); torch.cuda.synchronize(); dt = time.time() - t0
    r = plan.result()
    lf = plan.lf_walk(out, sarate=0)[:2]
    print("B3M_SPARSE_TIES=%-2s shards %d: %.1f ms, tie rounds %d on %d listed suffixes, LF %s" % (cap or "-", plan.host_tier, dt * 1e3, r["tie_rounds"], r["tie_elements"], lf), flush=True)
    res[cap] = out.clone()
    plan.close()
print("equal:", bool(torch.equal(res[""], res["0"])))
