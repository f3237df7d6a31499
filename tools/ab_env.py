"""A/B of the library's measurement switches on the C3 workload in ONE process (same plan, same text, same box):
python tools/ab_env.py [--size N] [--rate R] [--steps K] [--genome repeats] NAME=v1,v2,... [NAME2=...]   (the cross product is run; '-' = unset;
--genome 0.15 / 0.5: the genome-like text of bench.py's legs instead of uniform bases)
Prints per setting: ms per step (wall), the kernel-family times, and whether the output equals the first setting's; the first setting's output
is LF-verified on the device."""
import itertools, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bwtb3m_amd import benchutil, engine, synth
args = sys.argv[1:]
size, rate, steps, genome = None, 1024, 5, 0.0
while args and args[0].startswith("--"):
    k, v = args[0], args[1]; args = args[2:]
    if k == "--size": size = int(float(v))
    elif k == "--rate": rate = int(v)
    elif k == "--steps": steps = int(v)
    elif k == "--genome": genome = float(v)
names = [a.split("=")[0] for a in args]
vals = [a.split("=")[1].split(",") for a in args]
itype, d_text, n = synth.config_input_device("c3", size=size, device="cuda")
if genome > 0:
    d_text = synth.genome_like_device(n - 1, 7, "cuda", repeats=genome)
plan = engine.Plan(n, itype, isasamplingrate=rate)
buf = plan.text_buffer(); buf[:d_text.numel()].copy_(d_text[:buf.numel()])
out = torch.empty(n, dtype=torch.uint8, device="cuda")
ref = None
for combo in itertools.product(*vals):
    for k, v in zip(names, combo):
        if v == "-": os.environ.pop(k, None)
        else: os.environ[k] = v
    plan.run(buf, out); torch.cuda.synchronize()
    fam = {}
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.run(buf, out)
        benchutil.add_kernel_times(fam, plan.kernel_times())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    if ref is None:
        ref = out.clone()
        lf = plan.lf_walk(ref, sarate=0)[:2] if rate else None
        eq = "LF %s" % (lf,)
    else:
        eq = "equal to the first: %s" % bool(torch.equal(out, ref))
    print("%s: %.2f ms/step  %s  %s" % (" ".join("%s=%s" % kv for kv in zip(names, combo)), dt,
                                        {k: round(v["ms"] / steps, 2) for k, v in fam.items() if v["ms"] > 0}, eq), flush=True)
