#!/usr/bin/env python3
"""bench.py -- throughput of the BWT-construction hot path on MI355X.

A step = one pass of the hot path (packed text resident in HBM -> BWT symbols in HBM) over the
synthetic input of BASELINE.json's metric config (C3: 3 Gbase pacterm; falls back to the largest
size that fits).  Prints ONE JSON line (contract in the task statement), with the roofline of the
dominant kernel family (HIP events on the launch stream, collected inside the library) and a
CPU baseline (the oracle's restatement of balanced block merging on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# GPU time of every leg of this run, from the engine's own HIP events (the per-family times of each plan run) plus the wall time of the LF walks that
# verify: `config.gpu_seconds`.  A sampler that looks at the GPU every few seconds sees little of a run whose bulk is the CPU baseline; this figure says
# how long the GPU worked.
GPU_MS = {"sort": 0.0, "lf_walk": 0.0, "runs": 0}


def gpu_acc(plan):
    GPU_MS["sort"] += sum(k["ms"] for k in plan.kernel_times())
    GPU_MS["runs"] += 1


def lf_walk_timed(torch, plan, d_out):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = plan.lf_walk(d_out, sarate=0)
    torch.cuda.synchronize()
    GPU_MS["lf_walk"] += (time.perf_counter() - t0) * 1e3
    return r


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c3", help="c1|c2|c3 (BASELINE.json configs)")
    ap.add_argument("--size", type=float, default=0, help="override the number of bases/bytes")
    ap.add_argument("--blocksize", type=float, default=0, help="leaf block size in symbols (0 = engine default)")
    ap.add_argument("--blocks", type=int, default=0, help="number of leaf blocks (overrides --blocksize)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=float, default=400e6, help="symbols of the CPU-baseline sample")
    ap.add_argument("--mode", default="prefix", choices=["prefix", "blocks"],
                    help="multi-GPU: suffix-range shards (no exchange) or block merging with RCCL all-to-all")
    ap.add_argument("--no-also-blocks", dest="also_blocks", action="store_false",
                    help="N>1: skip the extra pass of the block-merge mode after the headline line")
    ap.add_argument("--blocks-timeout", type=int, default=90)
    ap.add_argument("--isasamplingrate", type=int, default=0, help="the option as given to bwtb3m (0 = the mode's default, 262144)")
    ap.add_argument("--bwtonly", type=int, default=0, help="the option as given to bwtb3m")
    ap.add_argument("--no-bwt-only", dest="bwt_only", action="store_false",
                    help="skip the extra leg that times a plan without (rank,pos) samples (`config.bwt_only`)")
    ap.add_argument("--no-genome-like", dest="genome_like", action="store_false",
                    help="skip the extra leg on a repeat-rich text of the same size (`config.genome_like`: what a real genome costs; never the headline)")
    ap.add_argument("--genome-repeats", type=float, default=0.5,
                    help="second repeat-rich leg (`config.genome_like_harsh`): fraction of the text in interspersed repeats, 0.5 = nearer to a human genome; 0 = skip")
    ap.add_argument("--no-check", dest="check", action="store_false",
                    help="skip the verification after the timed loop (default: the timed output is LF-walked on the device against the "
                         "text from the timed plan's own (rank,pos) samples, checkbwt semantics)")
    ap.add_argument("--verify", action="store_true", help="check the result against the oracle (small sizes only)")
    ap.add_argument("--copy-text", action="store_true", help="pass the text in a buffer of the caller's (copied into the plan every step) instead of the plan's own text buffer")
    ap.add_argument("--e2e-steps", type=int, default=3, help="host-memory runs (pinned input -> pinned BWT bytes) reported as `e2e`; 0 = skip")
    return ap.parse_args()


def cpu_baseline(buf, itype, sample, blocks):
    """oracle (kind 'port'): CPU restatement of balanced block merging on a prefix of the workload"""
    import numpy as np
    from oracle import oracle as O      # cpu_baseline leg: the only place bench.py touches oracle/
    # the GPU box exposes every host core; one GPU's share is 16 (task statement)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    if itype == "bytestream":
        sym = np.asarray(buf[:sample], dtype=np.uint8)
        sigma = 256
    else:
        nb = min(sample, (len(buf) - 2) * 4 + int(buf[-1]))
        data = np.asarray(buf[: (nb + 3) // 4], dtype=np.uint8)
        b = np.empty(len(data) * 4, dtype=np.uint8)
        b[0::4] = data >> 6; b[1::4] = (data >> 4) & 3; b[2::4] = (data >> 2) & 3; b[3::4] = data & 3
        b = b[:nb]
        sym = np.concatenate([b + 1, np.zeros(1, dtype=np.uint8)]).astype(np.uint8) if itype == "pacterm" else b
        sigma = 5 if itype == "pacterm" else 4
    B = max(blocks, cores)
    t0 = time.time()
    _, _, st = O.b3m(sym, B, cores, sigma)
    dt = time.time() - t0
    return {"value": len(sym) / dt / 1e9, "unit": "Gbase/s", "cores": cores, "kind": "port",
            "sample": "first %d symbols of the workload, %d leaf blocks, %d threads, %.1f s (leaf sort %.1f s, gap %.1f s, interleave %.1f s)"
                      % (len(sym), B, cores, dt, st[0], st[1], st[2])}


def verify_timed_result(torch, plan, d_text, d_out, n, itype):
    """After the timed loop: prove that what was timed IS the BWT.  The timed plan carries the (rank,pos) samples the shipped
    path emits (b3m_plan_sampling_rate), so its own output is LF-walked on the device against the text (every symbol, every chain
    closure: /root/reference/src/checkbwt.cpp:176-243); out[p0] is the terminator and the symbol histograms of the BWT, of the
    text (torch alone) and of the engine's .hist agree."""
    t0 = time.perf_counter()
    res0 = plan.result()
    from bwtb3m_amd import benchutil
    hist_text = benchutil.text_histogram(d_text, n, itype)          # torch only, independent of the engine
    assert plan.histogram().astype("int64").tolist() == hist_text, "bench check: the engine's symbol histogram (.hist) differs from the text's"
    m0, m1, _ = lf_walk_timed(torch, plan, d_out)
    assert (m0, m1) == (0, 0), "bench check: LF walk found %d symbol and %d chain mismatches" % (m0, m1)
    if itype == "pacterm":
        assert int(d_out[res0["p0"]]) == 0, "bench check: BWT[p0] is not the terminator"
    hb = torch.zeros(256, dtype=torch.int64, device=d_out.device)
    for o in range(0, n, 1 << 28):      # in slices: an int64 copy of 3 G symbols at once would be 24 GB
        hb += torch.bincount(d_out[o:o + (1 << 28)].to(torch.int64), minlength=256)
    assert hb.cpu().numpy().tolist() == hist_text, "bench check: symbol histogram of the BWT differs from the text's"
    return ("the timed output, LF-walked on the device against the text from the timed plan's own %d (rank,pos) samples over all "
            "%d symbols: 0 symbol / 0 chain mismatches; BWT[p0]=terminator; histogram(BWT)==histogram(text)==.hist; %.0f ms"
            % (res0["num_samples"], n, (time.perf_counter() - t0) * 1e3))


def time_bwt_only(torch, engine, d_text, d_out, n, itype, blocksize, steps):
    """the same workload through a plan without (rank,pos) samples (isasamplingrate=0: nothing b3m_compute_bwt ever creates): reported
    beside the headline, never instead of it; its output must equal the verified one"""
    plan0 = engine.Plan(n, itype, blocksize=blocksize, isasamplingrate=0)
    buf = plan0.text_buffer()
    buf[:d_text.numel()].copy_(d_text[:buf.numel()])
    d0 = torch.empty(n, dtype=torch.uint8, device=d_out.device)
    plan0.run(buf, d0)
    torch.cuda.synchronize()
    gpu_acc(plan0)
    t0 = time.perf_counter()
    for _ in range(steps):
        plan0.run(buf, d0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for _ in range(steps):
        gpu_acc(plan0)                       # (every step ran the same kernels: the last step's times, once per step)
    same = bool(torch.equal(d0, d_out))
    assert same, "bench: the isasamplingrate=0 plan's output differs from the shipped plan's"
    del buf
    plan0.close()
    nbases = n - 1 if itype == "pacterm" else n
    return {"ms_per_step": round(dt / steps * 1e3, 3), "value": round(nbases * steps / dt / 1e9, 4), "unit": "Gbase/s", "steps": steps,
            "equals_shipped_output": same, "what": "plan without (rank,pos) samples: BWT bytes only, not what b3m_compute_bwt runs"}


def time_genome_like(torch, engine, synth, n, itype, blocksize, plan_rate, repeats=0.15):
    """the same size and plan, on a text with the repeat structure of a genome (synth.genome_like_device: interspersed repeats at 3-10 %
    divergence, segmental duplications, microsatellites, a satellite array) instead of uniform bases: BASELINE.json's metric is quoted on
    uniform text, every real .pac is on this side.  LF-verified from the plan's own samples; two timed steps."""
    nb = n - 1 if itype == "pacterm" else n
    t0 = time.perf_counter()
    d_text = synth.genome_like_device(nb, 7, "cuda", repeats=repeats)
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t0
    plan = engine.Plan(n, itype, blocksize=blocksize, isasamplingrate=plan_rate)
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    plan.run(d_text, out)                                   # (the first run of a plan on repeat-rich text sorts twice: BWT only, then with the suffix array kept)
    torch.cuda.synchronize()
    gpu_acc(plan)
    steps = 2
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.run(d_text, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    for _ in range(steps):
        gpu_acc(plan)
    r = plan.result()
    m0, m1, _ = lf_walk_timed(torch, plan, out)
    assert (m0, m1) == (0, 0), "bench genome_like: LF walk found %d symbol and %d chain mismatches" % (m0, m1)
    fam = {k["name"]: round(k["ms"], 2) for k in plan.kernel_times() if k["ms"] > 0.05}
    plan.close()
    return {"ms_per_step": round(dt * 1e3, 2), "value": round(nb / dt / 1e9, 3), "unit": "Gbase/s", "steps": steps, "tie_rounds": r["tie_rounds"],
            "tie_elements": r["tie_elements"], "families_ms": fam, "lf_verified": True, "input_gen_s": round(gen_s, 2),
            "repeats": repeats,
            "what": "same size, same plan, genome-like repeat-rich text (bwtb3m_amd.synth.genome_like_device%s): not the metric's input"
                    % (", the harsher recipe: ~45 % of the positions in interspersed repeats at 8-16 % divergence" if repeats >= 0.3 else ": ~15 % of the positions planted")}


def measure_e2e(torch, plan, d_text, d_out, n, nbases, steps, itype):
    """input in host memory (page cache) -> BWT symbols in host memory, through PCIe both ways.
    Pinned host buffers; the BWT leaves the device in slices while later buckets are still being sorted.  Two forms of the
    hand-over: one byte per symbol, and (2-bit inputs) packed like the input, a quarter of the bytes (b3m_plan_set_packed_output)."""
    from bwtb3m_amd import engine
    h_text = d_text.cpu().pin_memory()
    out = {}
    # (a plan that runs sequential shards -- texts of 2^32 symbols and more, the host tier -- hands the BWT over one byte per symbol: b3m_plan_run_host refuses the packed form there)
    can_pack = itype in ("pac", "pacterm") and n < (1 << 32) and not getattr(plan, "host_tier", 0)
    for packed in ((False, True) if can_pack else (False,)):
        nbytes = (n + 3) // 4 if packed else n
        h_out = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        plan.run_host(h_text, h_out, packed=packed)    # warm-up (first touch of the pinned pages)
        gpu_acc(plan)
        acc = {"h2d_ms": 0.0, "construct_ms": 0.0, "d2h_tail_ms": 0.0, "total_ms": 0.0}
        streamed = 0
        per_step = []
        for _ in range(steps):
            t = plan.run_host(h_text, h_out, packed=packed)
            gpu_acc(plan)
            for k in acc:
                acc[k] += t[k]
            streamed = t["streamed_bytes"]
            # every step on its own (round 4 saw ONE host-to-host step of 56 ms, construct 41 instead of 26 ms, that left no record of which family grew): a
            # step whose construct time is a quarter above the fastest keeps the engine's family times
            per_step.append({"total_ms": round(t["total_ms"], 2), "construct_ms": round(t["construct_ms"], 2),
                             "families_ms": {k["name"]: round(k["ms"], 2) for k in plan.kernel_times() if k["ms"] > 0.05}})
        torch.cuda.synchronize()
        if packed:
            want = torch.cat([engine.Plan.pack_pac(c, itype == "pacterm") for c in torch.split(d_out, 1 << 28)]).cpu()   # (1 << 28 is a multiple of 4)
        else:
            want = d_out.cpu()
        ok = bool(torch.equal(h_out, want))
        assert ok, "bench e2e: the BWT that arrived in host memory differs from the device-resident result"
        res = {k: round(v / steps, 3) for k, v in acc.items()}
        fastest = min(p["construct_ms"] for p in per_step)
        res["per_step_total_ms"] = [p["total_ms"] for p in per_step]
        slow = [p for p in per_step if p["construct_ms"] > 1.25 * fastest]
        if slow:
            res["slow_steps"] = slow
        res.update({"value_e2e": round(nbases / (acc["total_ms"] / steps) / 1e6, 4), "unit": "Gbase/s", "steps": steps,
                    "bytes_in": int(h_text.numel()), "bytes_out": int(nbytes), "streamed_bytes": int(streamed), "equals_device_result": ok})
        out["packed" if packed else "bytes"] = res
        del h_out
    best = out.get("packed", out["bytes"])
    res = dict(best)
    res["hand_over"] = "2 bits per symbol (.pac layout)" if "packed" in out else "1 byte per symbol"
    if "packed" in out:
        res["one_byte_per_symbol"] = out["bytes"]
    res["what"] = ("pinned host text -> HBM (PCIe) -> construction -> BWT symbols in pinned host memory (PCIe), wall clock of "
                   "b3m_plan_run_host; RL file writing is separate (DESIGN.md 4)")
    return res


def benchutil_count_gpus():
    """GPUs of this machine without any HIP call: the KFD topology nodes with SIMDs (CPU nodes have simd_count 0); HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES narrow it"""
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for d in os.listdir(base):
            try:
                with open(os.path.join(base, d, "properties")) as f:
                    props = dict(l.split()[:2] for l in f if len(l.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                pass
    except OSError:
        return 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(a):
    """`python3 bench.py --gpus N` with no launcher around it (no WORLD_SIZE in the environment): this process starts the N ranks itself --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>`,
    the form the driver uses -- BEFORE anything here has touched a GPU (it never does: no HIP call, no torch.cuda.is_available()), lets
    rank 0's ONE JSON line through on the inherited stdout and returns the ranks' status.  (Do not profile an N > 1 run through this launcher: under
    `rocprofv3 ... -- python3 bench.py --gpus N` the profiler's library has initialised the GPU in this parent before it starts the ranks; profile one rank directly.)  One rank per GPU over RCCL; with
    B3M_BENCH_BACKEND=gloo the ranks share the GPUs that exist (rehearsal on a one-GPU box, collectives staged through host memory)."""
    import socket
    import subprocess
    ndev = benchutil_count_gpus()               # (from sysfs: nothing of HIP / torch.cuda is touched in the launcher, whatever torch's device count falls back to)
    if os.environ.get("B3M_BENCH_BACKEND", "nccl") == "nccl" and ndev > 0 and a.gpus > ndev:      # (0: the topology could not be read -- the ranks will say)
        sys.stderr.write("bench.py: --gpus %d but this machine shows %d GPU(s); RCCL needs one GPU per rank "
                         "(B3M_BENCH_BACKEND=gloo rehearses the N>1 flow with the ranks sharing a GPU)\n" % (a.gpus, ndev))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    import numpy as np
    import torch
    from bwtb3m_amd import benchutil, engine, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # (WORLD_SIZE=1 with B3M_BENCH_BACKEND set explicitly: the N > 1 code path with a process group of ONE rank -- every collective of the exchange mode runs,
    # through RCCL on a one-GPU box; a plain `python bench.py` or a launcher's WORLD_SIZE=1 alone is the one-GPU line)
    if a.gpus > 1 or world > 1 or ("WORLD_SIZE" in os.environ and os.environ.get("B3M_BENCH_BACKEND")):
        from bwtb3m_amd import distributed
        return distributed.bench_main(a, rank, local_rank, world)

    torch.cuda.set_device(0)
    itype, nsyms_cfg, _ = synth.CONFIGS[a.config]
    size = int(a.size) if a.size else None
    workload_note = ""
    # largest size that fits: the single-block workspace needs ~45 B/symbol
    free_b, _ = torch.cuda.mem_get_info()
    want = size or nsyms_cfg
    while want * 50 > free_b and want > 1_000_000:
        want //= 2
        workload_note = " (reduced: full size does not fit free HBM)"
    t0 = time.time()
    itype, d_text, n = synth.config_input_device(a.config, size=want, device="cuda")   # generated in HBM
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    blocksize = int(a.blocksize)
    if a.blocks > 0:
        blocksize = (n + a.blocks - 1) // a.blocks
    # the timed plan is the one b3m_compute_bwt creates for these options (same rule, b3m_plan_sampling_rate): BWT + (rank,pos) samples
    from bwtb3m_amd import _lib
    plan_rate = int(_lib.lib().b3m_plan_sampling_rate(a.isasamplingrate, a.bwtonly, n))
    plan = engine.Plan(n, itype, blocksize=blocksize, isasamplingrate=plan_rate)
    d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
    d_text_gen = d_text
    if not a.copy_text:
        # the input is resident in HBM in the plan's own text buffer (b3m_plan_text_buffer): the run uses it in place instead of copying
        # 0.75 GB into the plan first; --copy-text times the copying form (a text that lives in somebody else's buffer)
        buf = plan.text_buffer()
        buf[:d_text.numel()].copy_(d_text[:buf.numel()])
        d_text = buf
    for _ in range(a.warmup):
        plan.run(d_text, d_out)
        gpu_acc(plan)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fam = {}
    for _ in range(a.steps):
        plan.run(d_text, d_out)
        benchutil.add_kernel_times(fam, plan.kernel_times())
        gpu_acc(plan)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    d_text = d_text_gen                                   # (the view into the plan's buffer dies with the plan)
    res = plan.result()
    nbases = n - 1 if itype == "pacterm" else n
    value = nbases * a.steps / dt / 1e9
    B = res["num_blocks"]
    levels = res["levels"]
    roof = benchutil.dominant_roofline(fam, a.steps, dt, nbases, levels, n if B == 1 else 0, ROOT)
    out = {
        "metric": "Gbase/s BWT construction (3 Gbase pacterm) at 1/2/4/8 MI355X; % HBM roofline",
        "value": round(value, 4), "unit": "Gbase/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "%s: %d-symbol synthetic %s%s" % (a.config.upper(), n, itype, workload_note),
                   "blocks": B, "merge_levels": levels, "tie_rounds": res["tie_rounds"], "tie_elements": res["tie_elements"],
                   "parallelism": "1 GPU", "input_gen_s": round(gen_s, 2)},
        "roofline": roof,
    }
    out["config"]["isasamplingrate"] = {"option": a.isasamplingrate or (64 if a.bwtonly else 262144), "bwtonly": a.bwtonly, "plan": plan_rate,
                                        "samples": res["num_samples"],
                                        "what": "the plan b3m_compute_bwt creates for these options (b3m_plan_sampling_rate); the samples are emitted by the sort itself"}
    out["config"]["value_is"] = "device-resident: packed text in HBM -> BWT bytes + (rank,pos) samples in HBM (SURVEY 8d end-to-end figure: `e2e`)"
    out["config"]["text"] = ("in a buffer of the caller's, copied into the plan every step" if a.copy_text
                             else "in the plan's own text buffer (b3m_plan_text_buffer), used in place")
    if a.e2e_steps > 0:
        out["e2e"] = measure_e2e(torch, plan, d_text, d_out, n, nbases, a.e2e_steps, itype)
    if a.verify:
        from oracle import oracle as O
        sym = O.symbols_from_file_bytes(d_text.cpu().numpy(), itype)
        assert np.array_equal(d_out.cpu().numpy(), O.bwt(sym)), "bench --verify: BWT differs from the oracle"
        out["config"]["verified"] = True
    if a.check:
        out["config"]["checked"] = verify_timed_result(torch, plan, d_text, d_out, n, itype)
    plan.close()
    if a.bwt_only:
        out["config"]["bwt_only"] = time_bwt_only(torch, engine, d_text, d_out, n, itype, blocksize, max(1, min(a.steps, 10)))
    if a.genome_like and itype in ("pac", "pacterm") and n >= (1 << 22):
        out["config"]["genome_like"] = time_genome_like(torch, engine, synth, n, itype, blocksize, plan_rate)
        if a.genome_repeats > 0:
            out["config"]["genome_like_harsh"] = time_genome_like(torch, engine, synth, n, itype, blocksize, plan_rate, repeats=a.genome_repeats)
    if not a.no_cpu_baseline:
        sample = int(min(a.cpu_sample, n))
        # host copy of the prefix the CPU baseline works on (same bytes as the GPU workload)
        _, buf, _ = synth.config_input(a.config, size=min(want, sample + 64))
        out["cpu_baseline"] = cpu_baseline(buf, itype, sample, B)
    out["config"]["gpu_seconds"] = {"sort_kernels": round(GPU_MS["sort"] / 1e3, 3), "lf_walks": round(GPU_MS["lf_walk"] / 1e3, 3), "plan_runs": GPU_MS["runs"],
                                    "what": "GPU time of all legs of this run: the engine's HIP-event family times of every plan run + the wall time of the verifying LF walks"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
