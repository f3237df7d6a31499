import os, subprocess, sys, hashlib, numpy as np
sys.path.insert(0, "/root/repo")
from oracle import oracle as O
rng = np.random.default_rng(9)
nb = 900_001
b = (rng.random(nb) >= 0.97).astype(np.uint8) * rng.integers(1, 4, nb).astype(np.uint8)
buf = O.pac_encode(b)
os.makedirs("/tmp/es", exist_ok=True)
open("/tmp/es/t.pac", "wb").write(np.asarray(buf, dtype=np.uint8).tobytes())
outs = {}
for g, bo in ((1, 1), (3, 1), (5, 0), (1, 0)):
    out = "/tmp/es/g%d_%d.bwt" % (g, bo)
    env = dict(os.environ, B3M_DEBUG="1")
    if g > 1: env["B3M_DEVICES"] = ",".join(["0"] * g)
    r = subprocess.run(["/root/repo/bin/bwtb3m", "inputtype=pacterm", "bwtonly=%d" % bo, "verbose=1", "numgpus=%d" % g, "isasamplingrate=4096", "outputfilename=" + out, "/tmp/es/t.pac"], capture_output=True, text=True, env=env)
    print("numgpus=%d bwtonly=%d rc=%d" % (g, bo, r.returncode))
    print("\n".join(l for l in r.stderr.splitlines() if "shard" in l or "[W]" in l or "coded" in l or "LF walks" in l or "rror" in l)[:1500])
    outs[(g, bo)] = {s: hashlib.sha256(open(out[:-4] + s, "rb").read()).hexdigest()[:12] for s in (".bwt", ".hist", ".preisa", ".sa", ".isa") if os.path.exists(out[:-4] + s)}
print(outs[(1, 1)] == outs[(3, 1)], outs[(1, 0)] == outs[(5, 0)])
