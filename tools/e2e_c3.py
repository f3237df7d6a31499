"""End-to-end run of the CLI at the headline size on the GPU box (not a pytest: minutes of host work).
python tools/e2e_c3.py [nbases]"""
import hashlib, os, subprocess, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from bwtb3m_amd import engine, synth, _lib
import ctypes
def fhash(fn):
    h = hashlib.sha256()
    with open(fn, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()
nb = int(float(sys.argv[1])) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 3_000_000_000
tmp = "/tmp/b3m_e2e"; os.makedirs(tmp, exist_ok=True)
itype, d_text, n = synth.config_input_device("c3", size=nb, device="cuda")
buf = d_text.cpu().numpy()
pac = os.path.join(tmp, "c3.pac"); buf.tofile(pac)
print("pac file", os.path.getsize(pac), "bytes, n =", n, flush=True)
plan = engine.Plan(n, itype)
ref = plan.run(d_text); torch.cuda.synchronize()
href = hashlib.sha256(ref.cpu().numpy().tobytes()).hexdigest()
p0 = plan.result()["p0"]
plan.close(); del ref, d_text; torch.cuda.empty_cache()
for bwtonly in (1, 0, 0, 1):       # (the later runs are the warm ones: files in the page cache, driver initialised)
    out = os.path.join(tmp, "c3_%d.bwt" % bwtonly)
    t0 = time.time()
    r = subprocess.run(["/root/repo/bin/bwtb3m", "inputtype=pacterm", "bwtonly=%d" % bwtonly, "numthreads=16", "verbose=1", "outputfilename=" + out, pac],
                       capture_output=True, text=True)
    if "--mmap-ab" in sys.argv:        # the same run with a shared mapping of the output files instead of pwrite
        r2 = subprocess.run(["/root/repo/bin/bwtb3m", "inputtype=pacterm", "bwtonly=%d" % bwtonly, "numthreads=16", "verbose=1", "outputfilename=" + out + ".pw", pac],
                            capture_output=True, text=True, env=dict(os.environ, B3M_FILE_MMAP="1"))
        print("   with B3M_FILE_MMAP=1:", r2.stderr.strip().splitlines()[-2], flush=True)
        os.remove(out + ".pw")
    dt = time.time() - t0
    print("bwtb3m bwtonly=%d: rc %d, %.1f s wall" % (bwtonly, r.returncode, dt), flush=True)
    print("\n".join(r.stderr.strip().splitlines()[-12:]), flush=True)
    assert r.returncode == 0
    print({f: os.path.getsize(os.path.join(tmp, f)) for f in sorted(os.listdir(tmp)) if f.startswith("c3_%d" % bwtonly)}, flush=True)
    t0 = time.time()
    L = _lib.lib()
    sym = np.empty(n, dtype=np.uint8); nout = ctypes.c_uint64(0)
    rc = L.b3m_rl_read(out.encode(), sym.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(n), ctypes.byref(nout))
    assert rc == 0 and nout.value == n
    h = hashlib.sha256(sym.tobytes()).hexdigest()
    print("decode %.1f s; decoded BWT == device BWT: %s" % (time.time() - t0, h == href), flush=True)
    assert h == href
# numgpus=8 with the one device listed eight times (B3M_DEVICES: eight host threads, eight shard plans, peer copies inside one GPU): the files must be
# the ones numgpus=1 wrote, byte for byte
if "--no-numgpus" not in sys.argv:
    for bwtonly in (1, 0):
        out = os.path.join(tmp, "c3g_%d.bwt" % bwtonly)
        t0 = time.time()
        r = subprocess.run(["/root/repo/bin/bwtb3m", "inputtype=pacterm", "bwtonly=%d" % bwtonly, "numthreads=16", "verbose=1", "numgpus=8", "outputfilename=" + out, pac],
                           capture_output=True, text=True, env=dict(os.environ, B3M_DEVICES="0,0,0,0,0,0,0,0"))
        print("bwtb3m numgpus=8 (one device listed 8 times) bwtonly=%d: rc %d, %.1f s wall" % (bwtonly, r.returncode, time.time() - t0), flush=True)
        print("\n".join(r.stderr.strip().splitlines()[-12:]), flush=True)
        assert r.returncode == 0
        for suf in ((".bwt", ".hist", ".meta", ".preisa") if bwtonly else (".bwt", ".hist", ".meta", ".sa", ".isa")):
            a, b = os.path.join(tmp, "c3_%d%s" % (bwtonly, suf)), os.path.join(tmp, "c3g_%d%s" % (bwtonly, suf))
            same = fhash(a) == fhash(b)
            print("   %s: numgpus=8 file == numgpus=1 file: %s" % (suf, same), flush=True)
            assert same
            if suf != ".meta":
                os.remove(b)
# host tier: mem=8g is below what a merge tree needs at this size (7.25 B/symbol = 21.75 GB): sequential suffix-range shards, the BWT through pinned host
# memory; the files must be the unconstrained run's
if "--host-tier" in sys.argv:
    out = os.path.join(tmp, "c3t_1.bwt")
    t0 = time.time()
    r = subprocess.run(["/root/repo/bin/bwtb3m", "inputtype=pacterm", "bwtonly=1", "numthreads=16", "verbose=1", "mem=8g", "outputfilename=" + out, pac], capture_output=True, text=True)
    print("bwtb3m mem=8g bwtonly=1 (host tier): rc %d, %.1f s wall" % (r.returncode, time.time() - t0), flush=True)
    print("\n".join(r.stderr.strip().splitlines()[-12:]), flush=True)
    assert r.returncode == 0
    for suf in (".bwt", ".hist", ".meta", ".preisa"):
        a, b = os.path.join(tmp, "c3_1" + suf), os.path.join(tmp, "c3t_1" + suf)
        same = fhash(a) == fhash(b)
        print("   %s: mem=8g file == unconstrained file: %s" % (suf, same), flush=True)
        assert same
        os.remove(b)
t0 = time.time()
r = subprocess.run(["/root/repo/bin/bwtb3mtobwa", os.path.join(tmp, "c3_0.bwt"), os.path.join(tmp, "bwa.bwt"), os.path.join(tmp, "bwa.sa")], capture_output=True, text=True)
print("bwtb3mtobwa rc %d, %.1f s; %s" % (r.returncode, time.time() - t0, r.stderr.strip()[-300:]), flush=True)
print({f: os.path.getsize(os.path.join(tmp, f)) for f in ("bwa.bwt", "bwa.sa") if os.path.exists(os.path.join(tmp, f))})
