import sys, os, time, torch, numpy as np
sys.path.insert(0, '.')
from bwtb3m_amd import engine, synth
nb = int(float(sys.argv[1])); mode = sys.argv[2]
itype, d_text, n = synth.config_input_device("c3", size=nb, device="cuda")
def run(rate):
    plan = engine.Plan(n, itype, isasamplingrate=rate)
    out = plan.run(d_text).clone(); torch.cuda.synchronize(); plan.close(); return out
if mode == "save":
    torch.save(run(0).cpu(), "/tmp/bwt_old.pt"); print("saved", flush=True)
else:
    old = torch.load("/tmp/bwt_old.pt").cuda()
    for rate in (0, 65536):
        o = run(rate)
        bad = torch.nonzero(o != old).flatten()
        print("rate", rate, "mismatches", bad.numel(), bad[:16].tolist(), flush=True)
        if bad.numel():
            d = (bad[1:] - bad[:-1])
            print("  gaps", d[:16].tolist(), "vals", o[bad[:8]].tolist(), old[bad[:8]].tolist(), flush=True)
