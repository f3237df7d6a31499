"""One-off validation at the headline size: single block vs 3 blocks vs 8 suffix-range shards, LF-walk checks."""
import sys, time, hashlib, torch
sys.path.insert(0, "/root/repo")
from bwtb3m_amd import engine, synth
itype, d_text, n = synth.config_input_device("c3", device="cuda")
plan = engine.Plan(n, itype, isasamplingrate=1 << 20)
one = plan.run(d_text).clone(); torch.cuda.synchronize()
r1 = plan.result()
print("B=1: p0", r1["p0"], "LF", plan.lf_walk(one, sarate=0)[:2], flush=True)
plan.close()
off = 0; ok = True
for k in range(8):
    plan = engine.Plan(n, itype, shard=k, nshards=8)
    sl = plan.run(d_text); torch.cuda.synchronize()
    r = plan.result()
    same = bool(torch.equal(sl, one[off:off + sl.numel()])) and r["slice_offset"] == off
    ok &= same
    print("shard %d/8: offset %d len %d equal %s%s" % (k, off, sl.numel(), same, (" p0 ok %s" % (r["p0"] == r1["p0"])) if r["has_p0"] else ""), flush=True)
    off += sl.numel(); plan.close(); del sl
print("shards tile the BWT:", ok and off == n, flush=True)
plan = engine.Plan(n, itype, blocksize=(n + 2) // 3)
t0 = time.perf_counter(); m = plan.run(d_text); torch.cuda.synchronize()
print("B=%d levels %d: %.0f ms, equal to B=1: %s, p0 %s" % (plan.num_blocks, plan.result()["levels"], (time.perf_counter() - t0) * 1e3, bool(torch.equal(m, one)), plan.result()["p0"] == r1["p0"]), flush=True)
plan.close()
