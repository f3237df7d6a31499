"""One plan on a genome-like repeat-rich text (synth.genome_like_device), timed and LF-verified; for rocprofv3 --kernel-trace --stats:
python tools/genome_like_run.py [nbases] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bwtb3m_amd import engine, synth
nb = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
d_text = synth.genome_like_device(nb, 7, "cuda")
n = nb + 1
plan = engine.Plan(n, "pacterm", isasamplingrate=1024)
out = torch.empty(n, dtype=torch.uint8, device="cuda")
for it in range(steps + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan.run(d_text, out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    r = plan.result()
    print("run %d: %.1f ms  rounds %d listed %d  %s" % (it, dt * 1e3, r["tie_rounds"], r["tie_elements"],
                                                        {k["name"]: round(k["ms"], 1) for k in plan.kernel_times() if k["ms"] > 0.3}), flush=True)
print("LF", plan.lf_walk(out, sarate=0)[:2])
