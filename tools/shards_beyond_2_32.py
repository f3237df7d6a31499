import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, fuzzlib
from bwtb3m_amd import engine, synth
nb = (1 << 32) + 12344
itype, d_text, n = synth.config_input_device("c3", size=nb, device="cuda")
d_text = d_text.clone()
hi_byte = ((1 << 32) + 4000) // 4
d_text[hi_byte:hi_byte + 2000] = d_text[123_456_789:123_456_789 + 2000].clone()
pat = torch.tensor([0x1B, 0xE4, 0x27], dtype=torch.uint8, device="cuda")
d_text[700_000_000:700_000_000 + 1500] = pat.repeat(500)
d_text[hi_byte - 3000:hi_byte - 2500] = 0
seg = d_text[50_000_000:50_050_000].clone()
for c in range(40):
    at = 100_000_000 + c * 24_000_000 + 17 * c
    d_text[at:at + 50_000] = seg
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 6
t0 = time.time()
plans, slices, tied = fuzzlib.run_shards_lists(torch, engine, d_text, n, itype, ns, isarate=1 << 16)
print("shards %d: %.1f s, tied %d, rounds %s" % (ns, time.time() - t0, tied, [p.result()["tie_rounds"] for p in plans]), flush=True)
p0 = [p.result()["p0"] for p in plans if p.result()["has_p0"]]
for p in plans: p.close()
del plans
torch.cuda.empty_cache()
full = torch.cat(slices); del slices
plan = engine.Plan(n, itype, isasamplingrate=1 << 16)
one = plan.run(d_text); torch.cuda.synchronize()
print("equal to the one-plan BWT:", bool(torch.equal(full, one)), "p0", p0, plan.result()["p0"], "LF", plan.lf_walk(full, sarate=0)[:2], flush=True)
