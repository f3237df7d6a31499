"""Level 1 of the one-block sort takes 10.6 or 11.4 ms depending on the process (profiles/NOTES_r03.md).  Is it where the plan's arena lands in HBM?  Plans created one
after the other in ONE process (each destroyed before the next), then two alive at once: the family time of level 1 (msd_scatter) per plan.   python tools/l1_mode.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bwtb3m_amd import engine, synth, _lib
itype, d_text, n = synth.config_input_device("c3", device="cuda")
rate = int(_lib.lib().b3m_plan_sampling_rate(0, 0, n))
out = torch.empty(n, dtype=torch.uint8, device="cuda")
def fam(plan):
    for _ in range(4): plan.run(d_text, out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); plan.run(d_text, out); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    k = {x["name"]: x["ms"] for x in plan.kernel_times()}
    base = int(_lib.lib().b3m_plan_text_buffer(plan._h) or 0)
    return "arena %#x (mod 2^30: %#x, mod 2^21: %#x): step %.2f ms, level 1 %.2f, finish %.2f, level 0 %.2f" % (base, base & ((1 << 30) - 1), base & ((1 << 21) - 1), min(ts), k.get("msd_scatter", 0), k.get("bucket_finish", 0), k.get("msd_scatter_text", 0))
for i in range(8):
    p = engine.Plan(n, itype, isasamplingrate=rate)
    print("plan %d alone:" % i, fam(p), flush=True)
    p.close(); torch.cuda.empty_cache()
a = engine.Plan(n, itype, isasamplingrate=rate)
b = engine.Plan(n, itype, isasamplingrate=rate)
print("two alive, first :", fam(a), flush=True)
print("two alive, second:", fam(b), flush=True)
print("two alive, first again:", fam(a), flush=True)
