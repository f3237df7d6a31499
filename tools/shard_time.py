import sys, time, torch
sys.path.insert(0, "/root/repo")
from bwtb3m_amd import engine, synth
itype, d_text, n = synth.config_input_device("c3", device="cuda")
for ns in ([int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else (1, 2, 4, 8)):
    worst = 0
    for k in sorted({0, ns // 2, ns - 1}):
        plan = engine.Plan(n, itype, shard=k, nshards=ns)
        out = torch.empty(plan.slice_capacity, dtype=torch.uint8, device="cuda")
        plan.run(d_text, out); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): plan.run(d_text, out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        r = plan.result(); fam = {k["name"]: round(k["ms"], 2) for k in plan.kernel_times() if k["ms"] > 0.05}
        print("nshards", ns, "shard", k, "ms %.2f" % ms, "slice", r["slice_len"], "ws GB %.1f" % (plan.workspace_bytes / 1e9), fam, flush=True)
        worst = max(worst, ms)
        plan.close(); del out
    print("=> N=%d worst shard %.2f ms -> %.1f Gbase/s" % (ns, worst, (n - 1) / worst / 1e6), flush=True)
