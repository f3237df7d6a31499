"""Suffix-range shards of skewed text: borders by 32-base window fall inside level-0 digits.
85 % A in 4 shards and 70 % A in 12 shards (the two refusals of tools/fuzz_large.py in round 4), each shard timed, the slices compared with the
one-block BWT (LF-verified); then what the window test costs a shard of UNIFORM text (C3, shard 3 of 8: B3M_SHARD_KEYS=1 against the default).
python tools/skew_shards.py [n]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from bwtb3m_amd import engine, synth
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
N -= N % 4


def skewed(p_a, seed):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    parts = []
    rest = (1.0 - p_a) / 3.0
    for o in range(0, N, 1 << 28):
        m = min(1 << 28, N - o)
        u = torch.rand(m, device="cuda", generator=g)
        b = (u >= p_a).to(torch.uint8) + (u >= p_a + rest).to(torch.uint8) + (u >= p_a + 2 * rest).to(torch.uint8)
        bb = b.view(-1, 4)
        parts.append(((bb[:, 0] << 6) | (bb[:, 1] << 4) | (bb[:, 2] << 2) | bb[:, 3]).to(torch.uint8))
    return torch.cat(parts + [torch.zeros(66, dtype=torch.uint8, device="cuda")])


def shards(d_text, n, ns, which=None):
    outs, sizes, times = [], [], []
    for k in (range(ns) if which is None else which):
        plan = engine.Plan(n, "pacterm", shard=k, nshards=ns, isasamplingrate=1 << 16)
        out = torch.empty(plan.slice_capacity, dtype=torch.uint8, device="cuda")
        plan.run(d_text, out); torch.cuda.synchronize()
        t0 = time.perf_counter(); plan.run(d_text, out); torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
        r = plan.result()
        assert plan.tie_pending() == 0
        outs.append((r["slice_offset"], out[: r["slice_len"]].clone())); sizes.append(r["slice_len"])
        if which is not None:
            print("   shard %d of %d: %.2f ms" % (k, ns, times[-1]), {f["name"]: round(f["ms"], 2) for f in plan.kernel_times() if f["ms"] > 0.05}, flush=True)
        plan.close(); del out
    return outs, sizes, times


import subprocess, tempfile, hashlib
exe = "/root/repo/bin/bwtb3m"
for p_a, ns in ((0.85, 4), (0.70, 12)):
    d_text = skewed(p_a, 7)
    n = N + 1
    one = engine.Plan(n, "pacterm", isasamplingrate=1 << 16)
    ref = one.run(d_text); torch.cuda.synchronize()
    t0 = time.perf_counter(); ref = one.run(d_text); torch.cuda.synchronize(); t_one = (time.perf_counter() - t0) * 1e3
    lf = one.lf_walk(ref, sarate=0)[:2]
    print("%.0f %% A, n=%d: one block %.1f ms (LF %s)" % (p_a * 100, n, t_one, lf), flush=True)
    one.close(); del ref
    # the program with numgpus=N (one device listed N times): the shards' lists of tied suffixes are finished collectively; files against numgpus=1
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        inp = os.path.join(td, "t.pac")
        d_text.cpu().numpy()[: N // 4 + 2].tofile(inp)
        del d_text
        sums = {}
        for g in (1, ns):
            out = os.path.join(td, "g%d.bwt" % g)
            env = dict(os.environ, B3M_DEVICES=",".join(["0"] * g)) if g > 1 else dict(os.environ)
            r = subprocess.run([exe, "inputtype=pacterm", "bwtonly=1", "verbose=1", "numgpus=%d" % g, "outputfilename=" + out, inp], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            print("   numgpus=%d: rc %d" % (g, r.returncode)); print("   " + "\n   ".join(l for l in r.stderr.decode().splitlines() if l.startswith(("[V] const", "[V] %d suffix" % g, "[W]", "[b3m] shard"))), flush=True)
            sums[g] = {suf: hashlib.sha256(open(out[:-4] + suf, "rb").read()).hexdigest()[:16] for suf in (".bwt", ".hist", ".preisa")}
        print("   files of numgpus=%d equal those of numgpus=1: %s" % (ns, sums[1] == sums[ns]), flush=True)

itype, d_text, n = synth.config_input_device("c3", device="cuda")
for keys in (None, "1"):
    if keys is None: os.environ.pop("B3M_SHARD_KEYS", None)
    else: os.environ["B3M_SHARD_KEYS"] = keys
    print("C3 uniform, B3M_SHARD_KEYS=%s" % keys, flush=True)
    shards(d_text, n, 8, which=[0, 3])
