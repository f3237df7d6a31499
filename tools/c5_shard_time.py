"""usage: c5_shard_time.py [nshards] [shard] [config]
Time of ONE suffix-range shard of the C5 geometry (30 Gbase pacterm, 8 shards) on one MI355X: what each GPU of an
8-GPU node would do.  B3M_NO_FINE_SHARDS=1 gives the coarse level 0 (12 bits: 28.6 K-record final buckets)."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from bwtb3m_amd import engine, synth

nsh = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shard = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = sys.argv[3] if len(sys.argv) > 3 else "c5"
itype, d_text, n = synth.config_input_device(cfg, device="cuda")
plan = engine.Plan(n, itype, isasamplingrate=0, shard=shard, nshards=nsh)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sl = plan.run(d_text)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    r = plan.result()
    print("run %d: %.1f ms  slice %d symbols  -> whole job at %d GPUs: %.1f Gbase/s" % (it, dt * 1e3, r["slice_len"], nsh, (n - 1) / dt / 1e9), flush=True)
    print({k["name"]: round(k["ms"], 2) for k in plan.kernel_times() if k["ms"] > 0.05}, flush=True)
