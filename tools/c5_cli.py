"""The CLI with its DEFAULT options (bwtonly=0: X.bwt + X.sa + X.isa) at BASELINE configs[4]'s size, 30 Gbase pacterm, on one GPU (not a pytest: 7.5 GB of
input, ~17 GB of output, minutes of host work for the RL container above 2^32 symbols).  python tools/c5_cli.py [nbases]"""
import os, subprocess, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from bwtb3m_amd import synth
nb = int(float(sys.argv[1])) if len(sys.argv) > 1 else 30_000_000_000
tmp = "/tmp/b3m_c5"; os.makedirs(tmp, exist_ok=True)
itype, d_text, n = synth.config_input_device("c5", size=nb, device="cuda")
pac = os.path.join(tmp, "c5.pac")
with open(pac, "wb") as f:
    for o in range(0, d_text.numel(), 1 << 30):
        f.write(d_text[o:o + (1 << 30)].cpu().numpy().tobytes())
print("pac file", os.path.getsize(pac), "bytes, n =", n, flush=True)
hist = [0] * 5
del d_text; torch.cuda.empty_cache()
out = os.path.join(tmp, "c5.bwt")
t0 = time.time()
r = subprocess.run(["/root/repo/bin/bwtb3m", "inputtype=pacterm", "numthreads=16", "verbose=1", "outputfilename=" + out, pac], capture_output=True, text=True)
print("bwtb3m (default options: bwtonly=0) at n = %d: rc %d, %.1f s wall" % (n, r.returncode, time.time() - t0), flush=True)
print("\n".join(r.stderr.strip().splitlines()[-14:]), flush=True)
assert r.returncode == 0
sizes = {f: os.path.getsize(os.path.join(tmp, f)) for f in sorted(os.listdir(tmp)) if f.startswith("c5.") and f != "c5.pac"}
print(sizes, flush=True)
sa = np.fromfile(os.path.join(tmp, "c5.sa"), dtype=np.uint64, count=2)
isa = np.fromfile(os.path.join(tmp, "c5.isa"), dtype=np.uint64)
assert int(sa[0]) == 32 and int(sa[1]) == (n + 31) // 32 and sizes["c5.sa"] == 16 + 8 * int(sa[1])
assert int(isa[0]) == 262144 and int(isa[1]) == (n + 262143) // 262144 and len(isa) == 2 + int(isa[1])
print(".sa: rate 32, %d entries; .isa: rate 262144, %d entries; ISA[0] = %d (p0)" % (int(sa[1]), int(isa[1]), int(isa[2])), flush=True)
for f in sizes:
    os.remove(os.path.join(tmp, f))
os.remove(pac)
