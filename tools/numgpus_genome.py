"""bin/bwtb3m numgpus=N on the genome-like text of the bench's `config.genome_like` leg, all N shards on this one GPU (B3M_DEVICES=0,0,..): what the tie finish of a
repeat-rich text costs when the shards exchange their lists of tied suffixes (b3m_plan_list_*), and that the files equal the one-GPU run's.
python tools/numgpus_genome.py [nbases] [N]"""
import os, subprocess, sys, hashlib, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bwtb3m_amd import synth
nb = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = synth.genome_like_device(nb, 7, "cuda")
tmp = os.environ.get("TMPDIR", "/tmp")
inp = os.path.join(tmp, "genome_like.pac")
nbytes = nb // 4 + 1 + (1 if nb % 4 == 0 else 0)
d[:nbytes].cpu().numpy().tofile(inp)
del d; torch.cuda.empty_cache()
def fhash(fn):
    h = hashlib.sha256()
    with open(fn, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""): h.update(blk)
    return h.hexdigest()[:16]
hs = {}
for g in (1, N):
    out = os.path.join(tmp, "g%d.bwt" % g)
    env = dict(os.environ, B3M_DEVICES=",".join(["0"] * g)) if g > 1 else dict(os.environ)
    for rep in range(2):
        t0 = time.time()
        r = subprocess.run([os.path.join(ROOT, "bin", "bwtb3m"), "inputtype=pacterm", "bwtonly=1", "verbose=1", "numgpus=%d" % g, "outputfilename=" + out, inp], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        dt = time.time() - t0
        print("numgpus=%d run %d: rc %d, %.2f s" % (g, rep, r.returncode, dt), flush=True)
        for ln in r.stderr.decode().splitlines():
            if ln.startswith("[V]") or ln.startswith("[W]") or "rror" in ln: print("   ", ln, flush=True)
    hs[g] = {suf: fhash(out[:-4] + suf) for suf in (".bwt", ".hist", ".preisa")}
print("files equal:", hs[1] == hs[N], hs[1])
