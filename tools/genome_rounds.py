"""The genome-like text of bench.py's `config.genome_like` leg (same size, same plan) with the per-round log of the list-based rounds (B3M_DEBUG=2: one line per
round with its time and slots; =3 adds the census of group sizes).  python tools/genome_rounds.py [nbases] [debug level] [steps] [repeats: 0.15 | 0.5 = the harsher recipe]"""
import os, sys, time
dbg = sys.argv[2] if len(sys.argv) > 2 else "2"
import torch
sys.path.insert(0, "/root/repo")
from bwtb3m_amd import engine, synth, _lib
nb = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
repeats = float(sys.argv[4]) if len(sys.argv) > 4 else 0.15
n = nb + 1
d_text = synth.genome_like_device(nb, 7, "cuda", repeats=repeats)
rate = int(_lib.lib().b3m_plan_sampling_rate(0, 0, n))
plan = engine.Plan(n, "pacterm", isasamplingrate=rate)
out = torch.empty(n, dtype=torch.uint8, device="cuda")
plan.run(d_text, out); torch.cuda.synchronize()          # (the first run of a plan on repeat-rich text sorts twice)
for i in range(steps):
    if i == steps - 1:
        os.environ["B3M_DEBUG"] = dbg
    t0 = time.perf_counter(); plan.run(d_text, out); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    os.environ.pop("B3M_DEBUG", None)
    r = plan.result()
    print("run %d: %.1f ms  rounds %d listed %d " % (i, dt * 1e3, r["tie_rounds"], r["tie_elements"]), {k["name"]: round(k["ms"], 1) for k in plan.kernel_times() if k["ms"] > 0.05}, flush=True)
print("LF", plan.lf_walk(out, sarate=0)[:2])
