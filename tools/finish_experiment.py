"""Where k_finish_v3's time goes at C3: timing-only runs that leave the kernel after a phase (B3M_FINISH_EXPERIMENT: 1 records loaded, 2 counted, 3 scanned, 4 entries placed,
5 listed bins fixed; 0: the whole kernel).  The BWT of such a run is not valid.  Needs the experiment build (the exits cost the kernel 0.15 ms, so the product is built without
them):   make experiment && B3M_LIB=bwtb3m_amd/libb3m_exp.so python tools/finish_experiment.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bwtb3m_amd import engine, synth, _lib
itype, d_text, n = synth.config_input_device("c3", device="cuda")
rate = int(_lib.lib().b3m_plan_sampling_rate(0, 0, n))
plan = engine.Plan(n, itype, isasamplingrate=rate)
out = torch.empty(n, dtype=torch.uint8, device="cuda")
for f in ("0", "1", "2", "3", "4", "5", "0"):
    os.environ["B3M_FINISH_EXPERIMENT"] = f
    def step():
        t0 = time.perf_counter()
        try:
            plan.run(d_text, out)
        except Exception:          # (a run that left the finish early has no p0: the library says so after its last kernel; the wall clock is what is read here)
            pass
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3
    for _ in range(3): step()
    ts = [step() for _ in range(7)]
    print("leave after phase %s: step %.2f ms (best of 7)" % (f, min(ts)), flush=True)
