import os, sys
assert "libb3m_exp" in os.environ.get("B3M_LIB", ""), "timing-only switches exist in the experiment build alone: make experiment; B3M_LIB=bwtb3m_amd/libb3m_exp.so"
sys.path.insert(0, "/root/repo")
import torch
from bwtb3m_amd import engine, synth, _lib
nb = 3_000_000_000; n = nb + 1
d_text = synth.genome_like_device(nb, 7, "cuda")
rate = int(_lib.lib().b3m_plan_sampling_rate(0, 0, n))
plan = engine.Plan(n, "pacterm", isasamplingrate=rate)
out = torch.empty(n, dtype=torch.uint8, device="cuda")
plan.run(d_text, out); torch.cuda.synchronize()
os.environ["B3M_DEBUG"] = "2"
for f in ("0", "1", "2", "3", "4", "0"):
    os.environ["B3M_TT_EXPERIMENT"] = f
    print("== flags", f, file=sys.stderr, flush=True)
    try:
        plan.run(d_text, out); torch.cuda.synchronize()
    except Exception as ex:
        print("   (", str(ex)[:60], ")", file=sys.stderr, flush=True)
