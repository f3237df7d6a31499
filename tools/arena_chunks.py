"""The arena of a one-block plan as one hipMalloc or as physical pieces of c MB mapped into one address range (B3M_ARENA_CHUNK_MB): plans created one after the other in one
process, per plan: creation / destruction time, best step, level 1.   python tools/arena_chunks.py [chunk MB ...]   (PLANS=n plans per setting)"""
import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from bwtb3m_amd import engine, synth, _lib
itype, d_text, n = synth.config_input_device("c3", device="cuda")
rate = int(_lib.lib().b3m_plan_sampling_rate(0, 0, n))
out = torch.empty(n, dtype=torch.uint8, device="cuda")
for c in (sys.argv[1:] or ["", "256", "512", "1024"]):
    os.environ["B3M_ARENA_CHUNK_MB"] = c or "0"          # ("": one hipMalloc)
    res = []
    for i in range(int(os.environ.get("PLANS", "5"))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p = engine.Plan(n, itype, isasamplingrate=rate)
        torch.cuda.synchronize(); tc = (time.perf_counter() - t0) * 1e3
        for _ in range(4): p.run(d_text, out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); p.run(d_text, out); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        k = {x["name"]: x["ms"] for x in p.kernel_times()}
        t0 = time.perf_counter(); p.close(); torch.cuda.synchronize(); td = (time.perf_counter() - t0) * 1e3
        res.append("create %.0f ms / destroy %.0f ms / step %.2f / l1 %.2f" % (tc, td, min(ts), k["msd_scatter"]))
    print("chunk %-4s:" % (c or "-"), " | ".join(res), flush=True)
