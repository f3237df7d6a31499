"""(rank,pos) sampling rate of the plan: what the samples cost the sort (they are written by the finish kernels and copied to pinned host memory
inside the step) against what the LF walks over the result (the .sa / .isa stage, the checkbwt walk) gain from more, shorter chains.
python tools/rate_sweep.py [rates, comma list]   (C3)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bwtb3m_amd import benchutil, engine, synth
rates = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1024, 4096, 16384, 262144]
itype, d_text, n = synth.config_input_device("c3", device="cuda")
out = torch.empty(n, dtype=torch.uint8, device="cuda")
for rate in rates:
    plan = engine.Plan(n, itype, isasamplingrate=rate)
    buf = plan.text_buffer(); buf[:d_text.numel()].copy_(d_text[:buf.numel()])
    plan.run(buf, out); torch.cuda.synchronize()
    fam = {}; steps = 10
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.run(buf, out)
        benchutil.add_kernel_times(fam, plan.kernel_times())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    walks = []
    for sar in (0, 32):
        torch.cuda.synchronize(); t1 = time.perf_counter()
        m0, m1, _ = plan.lf_walk(out, sarate=sar)
        torch.cuda.synchronize(); walks.append((time.perf_counter() - t1) * 1e3)
        assert (m0, m1) == (0, 0)
    print("rate %7d: %.2f ms/step  finish %.2f  samples %d  LF walk (check only / with SA samples at 32) %.0f / %.0f ms" %
          (rate, dt, fam["bucket_finish"]["ms"] / steps, plan.result()["num_samples"], walks[0], walks[1]), flush=True)
    plan.close()
