"""GPU tests at the drop-in boundary: the bwtb3m command line and BwtMergeSort.computeBwt produce
the reference's file set (X.bwt RL container + X.hist + X.meta + X.preisa,
/root/reference/src/checkbwt.cpp:39-53, /root/reference/README.md:63-67) whose decoded content equals the
oracle; the committed golden fixtures are reproduced without the oracle."""
import ctypes
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "bwt_golden.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _decode(L, fn):
    n = ctypes.c_uint64()
    assert L.b3m_rl_length(fn.encode(), ctypes.byref(n)) == 0
    out = np.zeros(n.value, dtype=np.uint8)
    assert L.b3m_rl_read(fn.encode(), out.ctypes.data, n.value, ctypes.byref(n)) == 0, L.b3m_last_error()
    return out


@pytest.mark.parametrize("extra", [[], ["blocksize=40000"]])
def test_cli_pacterm_file_set(gpu, oracle, tmp_path, extra):
    from bwtb3m_amd import _lib
    L = _lib.lib()
    b = oracle.synth_bases(200_003, 21)
    buf = oracle.pac_encode(b)
    inp = tmp_path / "in.pac"
    inp.write_bytes(buf.tobytes())
    out = str(tmp_path / "x.pac.bwt")
    r = subprocess.run([os.path.join(ROOT, "bin", "bwtb3m"), "inputtype=pacterm", "bwtonly=1", "outputfilename=" + out, "verbose=1"] + extra + [str(inp)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == b"" and b"[M] " in r.stderr and b" runtime " in r.stderr     # /root/reference/src/bwtb3m.cpp:65
    s = oracle.symbols_from_file_bytes(buf, "pacterm")
    ref = oracle.bwt(s)
    assert np.array_equal(_decode(L, out), ref)
    prefix = out[:-4]
    for suf in (".hist", ".meta", ".preisa"):
        assert os.path.exists(prefix + suf), suf
    # .hist: count, then (key,value) 8-byte big-endian (bwtdecodeblock.cpp:150-173 reader)
    raw = open(prefix + ".hist", "rb").read()
    cnt = int.from_bytes(raw[:8], "big")
    hist = {int.from_bytes(raw[8 + 16 * i: 16 + 16 * i], "big"): int.from_bytes(raw[16 + 16 * i: 24 + 16 * i], "big") for i in range(cnt)}
    assert hist == {int(k): int(v) for k, v in zip(*np.unique(s, return_counts=True))}
    # .preisa: (rank,pos) u64 pairs, every 64th position when bwtonly=1 (ChangeLog:281)
    pre = np.fromfile(prefix + ".preisa", dtype=np.uint64).reshape(-1, 2)
    sa = oracle.sa_circular(s)
    isa = np.empty_like(sa); isa[sa] = np.arange(len(sa))
    assert len(pre) == (len(s) + 63) // 64
    assert np.array_equal(isa[pre[:, 1].astype(np.int64)], pre[:, 0].astype(np.int64))
    # the named consumer decodes the container
    d = subprocess.run([os.path.join(ROOT, "bin", "bwtb3mdecoderl"), out], stdout=subprocess.PIPE)
    assert d.returncode == 0 and d.stdout == ref.tobytes()


def _files(prefix):
    return {suf: open(prefix + suf, "rb").read() for suf in (".bwt", ".hist", ".meta", ".preisa", ".sa", ".isa") if os.path.exists(prefix + suf)}


@pytest.mark.parametrize("kind,ngpus,bwtonly", [("c2", 4, 1), ("c2", 3, 0), ("repeat", 4, 0), ("repeat", 2, 1), ("pac", 3, 1), ("pac", 2, 0)])
def test_cli_numgpus_writes_the_files_of_one_gpu(gpu, oracle, tmp_path, kind, ngpus, bwtonly):
    """b3m_compute_bwt honours num_gpus (CLI key numgpus=): one host thread per device, one suffix-range shard each, the slices gathered on the first
    device, which writes the files.  Here the same device is listed N times (B3M_DEVICES=0,0,..: N threads, N plans, peer copies within one GPU).
    C2 (5 Mbase pacterm) against the oracle; a repeat-rich text whose shards need the collective finish of long repeats (suffix array slices gathered
    by peer copies, b3m_plan_tie_*).  Every file must equal, byte for byte, what numgpus=1 writes (/root/reference/src/bwtb3m.cpp:48-50,62-63:
    numthreads= is the reference's knob behind the same call)."""
    from bwtb3m_amd import _lib, synth
    L = _lib.lib()
    itype = "pac" if kind == "pac" else "pacterm"
    if kind == "c2":
        buf = synth.pac_file_bytes(5_000_000, synth.SEED_BASE + 1)
    elif kind == "pac":
        buf = oracle.pac_encode(oracle.synth_repeat_bases(400_003, 31))          # circular text, no terminator
    else:
        b = oracle.synth_repeat_bases(700_001, 5)
        b[400_000:440_000] = b[100_000:140_000]            # a 40-kb copy: ties that no shard can finish from its own suffixes
        b[600_000:600_900] = 2                             # a homopolymer run
        buf = oracle.pac_encode(b)
    inp = tmp_path / "in.pac"
    inp.write_bytes(np.asarray(buf, dtype=np.uint8).tobytes())
    outs = {}
    for N in (1, ngpus):
        out = str(tmp_path / ("g%d.bwt" % N))
        env = dict(os.environ, B3M_DEVICES=",".join(["0"] * N)) if N > 1 else dict(os.environ)
        r = subprocess.run([os.path.join(ROOT, "bin", "bwtb3m"), "inputtype=" + itype, "bwtonly=%d" % bwtonly, "outputfilename=" + out, "verbose=1", "numgpus=%d" % N,
                            "isasamplingrate=2048", str(inp)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, r.stderr.decode()
        if N > 1:
            assert ("%d suffix-range shards" % N).encode() in r.stderr
            # the output stage is distributed too (round 5): every device codes the blocks of X.bwt that begin in its slice, and walks its share of the LF chains of X.sa
            assert b"X.bwt coded by" in r.stderr, r.stderr.decode()
            assert (b"LF walks on %d devices" % N in r.stderr) == (bwtonly == 0), r.stderr.decode()
            if kind == "repeat":
                assert b"tie finish" in r.stderr and b"(0 tied" not in r.stderr, r.stderr.decode()
        outs[N] = _files(out[:-4])
    assert set(outs[1]) == set(outs[ngpus]) == ({".bwt", ".hist", ".meta", ".preisa"} if bwtonly else {".bwt", ".hist", ".meta", ".sa", ".isa"})
    for suf in outs[1]:
        assert outs[1][suf] == outs[ngpus][suf], suf
    s = oracle.symbols_from_file_bytes(np.asarray(buf, dtype=np.uint8), itype)
    assert np.array_equal(_decode(L, str(tmp_path / ("g%d.bwt" % ngpus))), oracle.bwt(s))
    # more GPUs than the machine has, without the test switch: refused with a message; so is a byte alphabet (suffix-range shards are for pac / pacterm)
    r = subprocess.run([os.path.join(ROOT, "bin", "bwtb3m"), "inputtype=" + itype, "numgpus=64", "outputfilename=" + str(tmp_path / "no.bwt"), str(inp)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={k: v for k, v in os.environ.items() if k != "B3M_DEVICES"})
    assert r.returncode != 0 and b"numgpus=64" in r.stderr
    r = subprocess.run([os.path.join(ROOT, "bin", "bwtb3m"), "inputtype=bytestream", "numgpus=2", "outputfilename=" + str(tmp_path / "no.bwt"), str(inp)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, B3M_DEVICES="0,0"))
    assert r.returncode != 0 and b"numgpus > 1 constructs pac / pacterm" in r.stderr


def test_python_compute_bwt_bytestream(gpu, oracle, tmp_path):
    # BASELINE.json configs[0]: inputtype=bytestream bwtonly=1, through the mirrored interface
    from bwtb3m_amd import ArgInfo, BwtMergeSortOptions, BwtMergeSort, B3MError, _lib
    t = oracle.synth_bytes(1 << 20, 0xB3B3B3B3)
    inp = tmp_path / "in.bin"
    inp.write_bytes(t.tobytes())
    out = str(tmp_path / "o.bwt")
    opts = BwtMergeSortOptions(ArgInfo(["bwtb3m", "inputtype=bytestream", "bwtonly=1", "outputfilename=" + out, str(inp)]))
    res = BwtMergeSort.computeBwt(opts)
    assert res.bwtfn == out and os.path.exists(res.histfn) and os.path.exists(res.preisafn)
    assert np.array_equal(_decode(_lib.lib(), out), oracle.bwt(t))
    bad = BwtMergeSortOptions(ArgInfo(["bwtb3m", "inputtype=lz4", str(inp)]))
    with pytest.raises(B3MError):
        BwtMergeSort.computeBwt(bad)
    missing = BwtMergeSortOptions(ArgInfo(["bwtb3m", str(tmp_path / "nope")]))
    with pytest.raises(B3MError):
        BwtMergeSort.computeBwt(missing)


def test_golden_fixtures_without_oracle(gpu):
    """HIP path against tests/golden (committed vectors); the oracle is not imported here"""
    from bwtb3m_amd import engine, synth
    for s, l in GOLD["literal"].items():
        out, res, plan = engine.bwt_device(np.frombuffer(s.encode(), dtype=np.uint8), "bytestream")
        assert bytes(out).decode() == l
        plan.close()
    for n in (1024, 65536, 1 << 20):
        out, res, plan = engine.bwt_device(synth.byte_file_bytes(n, synth.SEED_BASE), "bytestream")
        assert sha(out) == GOLD["random_bytestream"][str(n)]
        plan.close()
    for nb in (1000, 65536, 5_000_000):
        g = GOLD["random_pacterm"][str(nb)]
        buf = synth.pac_file_bytes(nb, synth.SEED_BASE + 1)
        assert sha(buf) == g["pac_sha256"]
        for bs in (0, nb // 5 + 1):
            out, res, plan = engine.bwt_device(buf, "pacterm", blocksize=bs)
            assert sha(out) == g["sha256"] and res["p0"] == g["zero_at"]
            plan.close()
        if GOLD["random_pac"][str(nb)]:
            out, res, plan = engine.bwt_device(buf, "pac", blocksize=nb // 3 + 1)
            assert sha(out) == GOLD["random_pac"][str(nb)]
            plan.close()


def test_golden_intermediates_stage_level(gpu):
    import torch
    from bwtb3m_amd import engine, synth
    buf = synth.pac_file_bytes(65536, synth.SEED_BASE + 1)
    d = torch.from_numpy(buf).cuda()
    n = 65537
    leaves = []
    for k in range(4):
        g = GOLD["intermediate"]["leaf%d" % k]
        sa, bw, gt, p0 = engine.block_sort(d, n, "pacterm", g["a"], g["m"])
        w = gt.cpu().numpy().view(np.uint32)
        idx = np.arange(1, g["m"] + 1)
        bits = ((w[idx >> 5] >> (idx & 31)) & 1).astype(np.uint8)
        assert sha(sa.cpu().numpy().view(np.uint32)) == g["sa"] and sha(bw.cpu().numpy()) == g["bwt"] and p0 == g["p0"] and sha(bits) == g["gt"]
        leaves.append((g["a"], g["m"], sa, bw, gt, p0))
    for k in (0, 2):
        g = GOLD["intermediate"]["merge%d%d" % (k, k + 1)]
        a, m, sa, bw, gt, p0 = leaves[k]
        a2, m2, sa2, bw2, gt2, p02 = leaves[k + 1]
        chain_len = 1000
        nch = (m2 + chain_len - 1) // chain_len
        pos = np.array([(a2 + m2 - c * chain_len) % n for c in range(nch)], dtype=np.uint64)
        anc = engine.rank_of(d, n, "pacterm", sa, pos)
        assert int(anc[0]) == g["rank_end"]
        sr, G, gtP = engine.gap(d, n, "pacterm", a, m, m2, bw, p0, gt2, anc, chain_len)
        assert sha(G.cpu().numpy().view(np.uint32)) == g["G"]
        out, p0P = engine.interleave(sr, bw, bw2, p0)
        assert sha(out.cpu().numpy()) == g["bwt"] and p0P == g["p0"]


@pytest.mark.parametrize("itype,bs", [("pacterm", 0), ("pac", 0), ("pacterm", 70001)])
def test_lf_walk_checkbwt_and_sampled_sa(gpu, oracle, itype, bs):
    """SURVEY.md 8f rows 1 + 3: on-device checkbwt and the sampled SA stage"""
    torch = gpu
    from bwtb3m_amd import engine
    b = oracle.synth_repeat_bases(300_000, 77)
    buf = oracle.pac_encode(b)
    s = oracle.symbols_from_file_bytes(buf, itype)
    sa = oracle.sa_circular(s)
    n = len(s)
    d = torch.from_numpy(buf).cuda()
    plan = engine.Plan(n, itype, blocksize=bs, isasamplingrate=256)
    out = plan.run(d)
    assert np.array_equal(out.cpu().numpy(), oracle.bwt(s))
    m0, m1, ssa = plan.lf_walk(out, sarate=32)
    assert (m0, m1) == (0, 0)
    assert np.array_equal(ssa.astype(np.int64), sa[::32])
    # a damaged BWT must be caught: swap two different symbols
    bad = out.clone()
    h = out.cpu().numpy()
    term = 0 if itype == "pacterm" else 255          # never move the terminator entry
    i = next(k for k in range(1000, n) if h[k] != term)
    j = next(k for k in range(i + 1, n) if h[k] != h[i] and h[k] != term)
    bad[i], bad[j] = out[j], out[i]
    m0, m1, _ = plan.lf_walk(bad, sarate=0)
    assert m0 + m1 > 0
    plan.close()


@pytest.mark.parametrize("itype", ["pacterm", "pac"])
@pytest.mark.parametrize("sbshift", [1, 3, 7])
def test_lf_walk_counts_in_64_bits_across_superblocks(gpu, oracle, itype, sbshift, monkeypatch):
    """the LF walk's rank dictionary counts in 64 bits (a table of counts per superblock of 2^k lines + 32-bit line headers that count from the superblock's
    start): what lets b3m_plan_lf_walk / .sa / .isa work when a symbol occurs 2^32 times and more (30 Gbase).  B3M_RANK_SB_SHIFT makes the superblocks
    2, 8 and 128 lines long, so this 300-kbase text spans 782 / 196 / 13 of them; the sampled SA must equal the oracle's.  Also the walk as a stage of
    its own (b3m_lf_walk: BWT, p0, histogram and samples handed in, as the multi-GPU path of b3m_compute_bwt does)."""
    torch = gpu
    from bwtb3m_amd import engine
    monkeypatch.setenv("B3M_RANK_SB_SHIFT", str(sbshift))
    b = oracle.synth_repeat_bases(300_011, 78)
    buf = oracle.pac_encode(b)
    s = oracle.symbols_from_file_bytes(buf, itype)
    sa = oracle.sa_circular(s)
    n = len(s)
    d = torch.from_numpy(np.concatenate([buf, np.zeros(16, dtype=np.uint8)])).cuda()
    plan = engine.Plan(n, itype, isasamplingrate=512)
    out = plan.run(d)
    m0, m1, ssa = plan.lf_walk(out, sarate=16)
    assert (m0, m1) == (0, 0) and np.array_equal(ssa.astype(np.int64), sa[::16])
    res = plan.result()
    m0, m1, ssa2 = engine.lf_walk(d, n, itype, out, res["p0"], plan.histogram(), plan.samples(), sarate=16)
    assert (m0, m1) == (0, 0) and np.array_equal(ssa2, ssa)
    bad = out.clone()
    h = out.cpu().numpy()
    term = 0 if itype == "pacterm" else 255
    i = next(k for k in range(200_000, n) if h[k] != term)
    j = next(k for k in range(i + 1, n) if h[k] != h[i] and h[k] != term)
    bad[i], bad[j] = out[j], out[i]
    m0, m1, _ = engine.lf_walk(d, n, itype, bad, res["p0"], plan.histogram(), plan.samples(), sarate=0)
    assert m0 + m1 > 0
    plan.close()


def test_cli_bwtonly0_writes_sa_isa(gpu, oracle, tmp_path):
    b = oracle.synth_bases(400_001, 88)
    buf = oracle.pac_encode(b)
    inp = tmp_path / "g.pac"
    inp.write_bytes(buf.tobytes())
    out = str(tmp_path / "g.pac.bwt")
    # the README's indexing call: no bwtonly => sampled SA and ISA (/root/reference/README.md:106-108)
    r = subprocess.run([os.path.join(ROOT, "bin", "bwtb3m"), "inputtype=pacterm", "outputfilename=" + out, "isasamplingrate=4096", str(inp)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    s = oracle.symbols_from_file_bytes(buf, "pacterm")
    sa = oracle.sa_circular(s)
    isa = np.empty_like(sa); isa[sa] = np.arange(len(sa))
    prefix = out[:-4]
    assert not os.path.exists(prefix + ".preisa")
    fsa = np.fromfile(prefix + ".sa", dtype=np.uint64)        # rate, count, values (sasubsample.cpp:34-57)
    assert fsa[0] == 32 and fsa[1] == len(fsa) - 2 and np.array_equal(fsa[2:].astype(np.int64), sa[::32])
    fisa = np.fromfile(prefix + ".isa", dtype=np.uint64)
    assert fisa[0] == 4096 and fisa[1] == len(fisa) - 2 and np.array_equal(fisa[2:].astype(np.int64), isa[::4096])


def test_readme_bwa_index_pipeline(gpu, oracle, tmp_path):
    """the reference's indexing script (/root/reference/README.md:106-111) minus bwa itself:
    bwtb3m inputtype=pacterm -> bwtb3mtobwa"""
    b = oracle.synth_bases(300_007, 91)
    buf = oracle.pac_encode(b)
    pac = tmp_path / "ref.fa.pac"
    pac.write_bytes(buf.tobytes())
    out = str(pac) + ".bwt"
    r = subprocess.run([os.path.join(ROOT, "bin", "bwtb3m"), "inputtype=pacterm", "mem=2g", "outputfilename=" + out, str(pac)], stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    r = subprocess.run([os.path.join(ROOT, "bin", "bwtb3mtobwa"), out, str(tmp_path / "ref.fa.bwt"), str(tmp_path / "ref.fa.sa")], stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    s = oracle.symbols_from_file_bytes(buf, "pacterm")
    sa = oracle.sa_circular(s)
    bw = oracle.bwt(s)
    primary = int(np.nonzero(sa == 0)[0][0])
    raw = np.fromfile(str(tmp_path / "ref.fa.bwt"), dtype=np.uint8)
    assert raw[:8].view(np.uint64)[0] == primary
    words = raw[40:].view(np.uint32)
    sym = ((words[:, None] >> np.arange(30, -2, -2, dtype=np.uint32)[None, :]) & 3).astype(np.uint8).reshape(-1)[: len(s) - 1]
    assert np.array_equal(sym, np.delete(bw, primary) - 1)
    fsa = np.fromfile(str(tmp_path / "ref.fa.sa"), dtype=np.uint64)
    assert fsa[5] == 32 and fsa[6] == len(s) - 1 and np.array_equal(fsa[7:].astype(np.int64), sa[32::32])


@pytest.mark.parametrize("count,nbits,kind", [(1000, 20, "uniform"), (20_000_000, 31, "uniform"), (20_000_000, 29, "uniform"),
                                               (20_000_000, 31, "skewed"), (9_000_000, 24, "uniform")])
def test_sort_u32_stage_entry(gpu, count, nbits, kind):
    # b3m_sort_u32 (row a9 helper): large lists take the MSD partition + LDS histogram path, skewed ones
    # (a final bucket of >= 2^16 keys) and small ones the LSD passes; either way the result is the sorted list
    torch = gpu
    import ctypes
    from bwtb3m_amd import engine as eng
    g = torch.Generator(device="cuda"); g.manual_seed(count + nbits)
    keys = torch.randint(0, 1 << nbits, (count,), device="cuda", generator=g, dtype=torch.int64)
    if kind == "skewed":
        keys = torch.where(torch.rand(count, device="cuda", generator=g) < 0.5, torch.full_like(keys, 12345), keys)
    k32 = keys.to(torch.int32)            # values < 2^31: same bits as u32
    out = torch.empty_like(k32)
    eng.check(eng.lib().b3m_sort_u32(eng._ptr(k32), ctypes.c_uint64(count), nbits, eng._ptr(out), eng._stream()))
    torch.cuda.synchronize()
    assert torch.equal(out, torch.sort(k32).values)


def test_cli_bwtonly0_bytestream_writes_sa_isa(gpu, oracle, tmp_path):
    # byte alphabets have no 2-bit rank dictionary: the sampled SA/ISA stage walks the LF array
    # (one stable sort of the BWT by symbol) and verifies every symbol against the text on the way
    t = oracle.synth_bytes(200_000, 93).copy()
    t[50_000:90_000] = t[10_000:50_000]          # a long repeat
    inp = tmp_path / "t.bin"
    inp.write_bytes(t.tobytes())
    out = str(tmp_path / "t.bin.bwt")
    r = subprocess.run([os.path.join(ROOT, "bin", "bwtb3m"), "inputtype=bytestream", "outputfilename=" + out, "isasamplingrate=2048", "sasamplingrate=16", str(inp)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    sa = oracle.sa_circular(t)
    isa = np.empty_like(sa); isa[sa] = np.arange(len(sa))
    prefix = out[:-4]
    assert not os.path.exists(prefix + ".preisa")
    fsa = np.fromfile(prefix + ".sa", dtype=np.uint64)
    assert fsa[0] == 16 and fsa[1] == len(fsa) - 2 and np.array_equal(fsa[2:].astype(np.int64), sa[::16])
    fisa = np.fromfile(prefix + ".isa", dtype=np.uint64)
    assert fisa[0] == 2048 and fisa[1] == len(fisa) - 2 and np.array_equal(fisa[2:].astype(np.int64), isa[::2048])


def test_lf_walk_bytestream_catches_corruption(gpu, oracle):
    from bwtb3m_amd import engine
    t = oracle.synth_bytes(300_000, 94)
    out, res, plan = engine.bwt_device(t, "bytestream", isasamplingrate=4096)
    import torch
    d = torch.from_numpy(np.ascontiguousarray(out)).cuda()
    m0, m1, sa = plan.lf_walk(d, sarate=64)
    assert (m0, m1) == (0, 0)
    ref = oracle.sa_circular(t)
    assert np.array_equal(np.asarray(sa, dtype=np.int64), ref[::64])
    d[12345] = (int(d[12345]) + 1) % 256
    m0, m1, _ = plan.lf_walk(d, sarate=0)
    assert m0 + m1 > 0
    plan.close()


def test_bench_line_contract(gpu):
    # bench.py prints ONE JSON line with the metric of BASELINE.json plus the `roofline` and `cpu_baseline` objects
    import json, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", "2e7", "--steps", "2", "--warmup", "1", "--cpu-sample", "2e6"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert j["metric"] == base["metric"] and j["unit"] == "Gbase/s" and j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1
    assert j["higher_is_better"] is True and j["vs_baseline"] is None and j["data"] == "synthetic" and j["value"] > 0
    assert abs(j["value"] - 2e7 * 1e-9 / (j["ms_per_step"] * 1e-3)) < 0.02 * j["value"]
    assert "workload" in j["config"] and "model" not in j["config"]
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    # what is timed is the plan b3m_compute_bwt creates (samples at b3m_plan_sampling_rate), verified from its own samples;
    # the BWT-only plan is reported beside it
    sr = j["config"]["isasamplingrate"]
    assert sr["option"] == 262144 and sr["plan"] == 1024 and sr["samples"] == (20_000_001 + 1023) // 1024      # (a text this short: 1024; C3: 16384)
    from bwtb3m_amd import _lib
    rate = _lib.lib().b3m_plan_sampling_rate
    assert [int(rate(0, 0, n)) for n in (20_000_001, 600_000_000, 3_000_000_001)] == [1024, 4096, 16384] and int(rate(0, 1, 3_000_000_001)) == 64
    assert int(rate(1000, 0, 3_000_000_001)) == 1000 and int(rate(8192, 0, 3_000_000_001)) == 4096
    assert "0 symbol / 0 chain mismatches" in j["config"]["checked"]
    assert j["config"]["bwt_only"]["equals_shipped_output"] is True and j["config"]["bwt_only"]["ms_per_step"] > 0


def test_mem_target_that_cannot_be_met_is_said_not_hidden(gpu, oracle, tmp_path):
    """mem= below the footprint of the smallest blocks (/root/reference/README.md:42: the reference's memory target): the plan says so
    (b3m_plan_warning), b3m_compute_bwt prints it as a [W] line whatever verbose= says, and the result is still the BWT"""
    from bwtb3m_amd import engine
    torch = gpu
    b = oracle.synth_bases(3_000_000, 5)
    buf = oracle.pac_encode(b)
    s = oracle.symbols_from_file_bytes(buf, "pacterm")
    ok = engine.Plan(len(s), "pacterm", mem=1 << 30)
    assert ok.warning == "" and ok.workspace_bytes <= 1 << 30
    ok.close()
    plan = engine.Plan(len(s), "pacterm", mem=8 << 20)                # 8 MB for 3 M symbols: below the 7.25 B/symbol that do not shrink
    assert "cannot be met" in plan.warning and plan.workspace_bytes > 8 << 20 and plan.num_blocks > 1
    out = plan.run(torch.from_numpy(buf).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), oracle.bwt(s))
    plan.close()
    inp = tmp_path / "in.pac"
    inp.write_bytes(buf.tobytes())
    r = subprocess.run([os.path.join(ROOT, "bin", "bwtb3m"), "inputtype=pacterm", "bwtonly=1", "mem=8m", "outputfilename=" + str(tmp_path / "x.bwt"), str(inp)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"[W] mem=8388608 cannot be met" in r.stderr, r.stderr.decode()


@pytest.mark.parametrize("kind", ["uniform", "repeat", "pac"])
def test_host_tier_sequential_shards_against_the_oracle(gpu, oracle, kind, monkeypatch):
    """HOST TIER (role of the reference's mem= / tmp files, /root/reference/README.md:42,46): a plan whose memory target no merge tree can meet sorts the text as
    sequential suffix-range shards; b3m_plan_run_host sends every slice of the BWT to pinned host memory while the next shard sorts (two slice buffers in HBM).
    Forced here on a text the oracle can sort (B3M_FORCE_SPILL=5); long repeats are finished shard by shard (key extension to the end)."""
    torch = gpu
    from bwtb3m_amd import engine
    monkeypatch.setenv("B3M_FORCE_SPILL", "5")
    itype = "pac" if kind == "pac" else "pacterm"
    if kind == "uniform":
        b = oracle.synth_bases(700_003, 9)
    else:
        b = oracle.synth_repeat_bases(700_003, 9)
        b[500_000:512_000] = b[40_000:52_000]                 # a 12-kb copy
        b[300_000:300_700] = 1
    buf = oracle.pac_encode(b)
    s = oracle.symbols_from_file_bytes(buf, itype)
    ref = oracle.bwt(s)
    sa = oracle.sa_circular(s)
    n = len(s)
    plan = engine.Plan(n, itype, isasamplingrate=1024)
    assert plan.host_tier == 5 and plan.num_blocks == 1
    d = torch.from_numpy(np.concatenate([buf, np.zeros(16, dtype=np.uint8)])).cuda()
    out = plan.run(d)                                          # device-resident: the caller's n bytes
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref)
    h_text = torch.from_numpy(np.concatenate([buf, np.zeros(64, dtype=np.uint8)])).pin_memory()
    h_out = torch.zeros(n, dtype=torch.uint8).pin_memory()
    t = plan.run_host(h_text, h_out)
    assert np.array_equal(h_out.numpy(), ref) and t["streamed_bytes"] == n
    r = plan.result()
    assert r["p0"] == int(np.nonzero(sa == 0)[0][0])
    smp = plan.samples()
    isa = np.empty_like(sa); isa[sa] = np.arange(n)
    assert len(smp) == (n + 1023) // 1024 and np.array_equal(isa[smp[:, 1].astype(np.int64)], smp[:, 0].astype(np.int64))
    assert plan.lf_walk(out, sarate=0)[:2] == (0, 0)
    plan.close()


@pytest.mark.parametrize("how", ["numgpus", "mem"])
def test_cli_refusing_shards_fall_back_with_a_warning(gpu, oracle, tmp_path, how):
    """a suffix-range shard that refuses its range must not end `bwtb3m numgpus=N` or `bwtb3m mem=...` (the reference never fails because of
    mem= or its thread count, /root/reference/src/bwtb3m.cpp:48-55).  85 % A with whole-digit borders forced (B3M_SHARD_KEYS=0: one level-0 digit holds 41 % of the
    suffixes; by default the borders fall inside the digit and nothing refuses -- second half): the job falls back to one device / to block merging with a [W] line
    and writes the files of an unconstrained run."""
    from bwtb3m_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(5)
    nb = 8_000_001          # (large enough that the plan's fixed slack of 2^20 suffixes does not hide the skew: a shard of four may hold 3.2 M, the digit holds 3.3 M)
    b = (rng.random(nb) >= 0.85).astype(np.uint8) * rng.integers(1, 4, nb).astype(np.uint8)
    buf = oracle.pac_encode(b)
    inp = tmp_path / "skew.pac"
    inp.write_bytes(np.asarray(buf, dtype=np.uint8).tobytes())
    exe = os.path.join(ROOT, "bin", "bwtb3m")
    common = ["inputtype=pacterm", "bwtonly=1", "verbose=1", "isasamplingrate=2048"]
    free = str(tmp_path / "free.bwt")
    r = subprocess.run([exe] + common + ["outputfilename=" + free, str(inp)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    want = _files(free[:-4])
    s = oracle.symbols_from_file_bytes(np.asarray(buf, dtype=np.uint8), "pacterm")
    ref = oracle.bwt(s)
    assert np.array_equal(_decode(L, free), ref)
    extra, env = (["numgpus=4"], dict(os.environ, B3M_DEVICES="0,0,0,0")) if how == "numgpus" else (["mem=64m"], dict(os.environ, B3M_FORCE_SPILL="4"))
    for keys in ("0", None):
        e = dict(env)
        if keys is not None:
            e["B3M_SHARD_KEYS"] = keys
        out = str(tmp_path / ("k%s.bwt" % keys))
        r = subprocess.run([exe] + common + extra + ["outputfilename=" + out, str(inp)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
        assert r.returncode == 0, r.stderr.decode()
        fell = b"[W]" in r.stderr and b"suffix-range shard larger than the workspace" in r.stderr
        assert fell == (keys == "0"), r.stderr.decode()
        if fell:
            assert (b"constructing on one device instead" if how == "numgpus" else b"constructing by block merging") in r.stderr
        got = _files(out[:-4])
        assert set(got) == set(want)
        for suf in want:
            if suf in (".bwt", ".hist") or not (fell and how == "mem"):          # (.meta names the blocks of the tree that ran; its .preisa is checked by its own tests)
                assert got[suf] == want[suf], (keys, suf)
        assert np.array_equal(_decode(L, out), ref)


@pytest.mark.parametrize("ngpus,bwtonly", [(3, 1), (5, 0)])
def test_cli_numgpus_with_an_empty_shard(gpu, oracle, tmp_path, ngpus, bwtonly):
    """97 % A: the all-A window holds more than a shard's share, so two borders coincide and a shard is EMPTY; the tie finish over the shards' lists, the split coder (one block:
    one device codes it, the others have no block that begins in their slice) and the split LF walks must cope, and the files are those of one GPU."""
    from bwtb3m_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(9)
    nb = 900_001
    b = (rng.random(nb) >= 0.97).astype(np.uint8) * rng.integers(1, 4, nb).astype(np.uint8)
    buf = oracle.pac_encode(b)
    inp = tmp_path / "a.pac"
    inp.write_bytes(np.asarray(buf, dtype=np.uint8).tobytes())
    outs = {}
    for N in (1, ngpus):
        out = str(tmp_path / ("g%d.bwt" % N))
        env = dict(os.environ, B3M_DEBUG="1")
        if N > 1:
            env["B3M_DEVICES"] = ",".join(["0"] * N)
        r = subprocess.run([os.path.join(ROOT, "bin", "bwtb3m"), "inputtype=pacterm", "bwtonly=%d" % bwtonly, "verbose=1", "numgpus=%d" % N, "isasamplingrate=4096",
                            "outputfilename=" + out, str(inp)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, r.stderr.decode()
        if N > 1:
            assert b"0 suffixes behind" in r.stderr and b"X.bwt coded by" in r.stderr, r.stderr.decode()
        outs[N] = _files(out[:-4])
    assert outs[1] == outs[ngpus]
    s = oracle.symbols_from_file_bytes(np.asarray(buf, dtype=np.uint8), "pacterm")
    assert np.array_equal(_decode(L, str(tmp_path / ("g%d.bwt" % ngpus))), oracle.bwt(s))


def _long_repeat_bases(oracle, nb, seed):
    """random bases with planted long repeats of both kinds: families above 32 copies (their suffixes reach the list-based rounds: a 16-mer above 32 records) -- a 2-kb
    segment 40 times, a tandem array, a homopolymer run -- and two-copy repeats (finished by the record sort's direct comparisons) that the copies of a family run into."""
    b = oracle.synth_bases(nb, seed)
    seg = b[10_000:12_000].copy()
    tail = b[20_000:26_000].copy()
    at = 60_000
    for c in range(40):
        b[at:at + 2_000] = seg
        if c in (7, 23): b[at + 2_000:at + 8_000] = tail       # two of the copies go on with the same 6 kb: the family's suffixes near its end look into a two-copy repeat
        at += 2_000 + (6_000 if c in (7, 23) else 0) + 37 + 13 * c
    assert at < 330_000
    b[400_000:403_000] = np.tile(np.array([0, 1, 2, 3, 3, 1, 0, 2, 2, 1, 3, 0], dtype=b.dtype), 250)      # tandem array, unit 12
    b[450_000:451_500] = 3
    b[520_000:535_000] = b[340_000:355_000]                   # a 15-kb two-copy repeat
    return b


@pytest.mark.gpu
@pytest.mark.parametrize("itype", ["pacterm", "pac"])
@pytest.mark.parametrize("cap", ["", "0", "300"])
def test_sequential_shards_finish_long_repeats_by_doubling_on_their_list(gpu, oracle, itype, cap, monkeypatch):
    """Row a6 for plans that run suffix-range shards one after the other (the host tier; texts of 2^32 symbols and more): every shard stops after four key-extension
    rounds and hands over what is left tied; the plan finishes the union by prefix doubling ON THAT LIST (tie_rounds_sparse: the rank of the suffix d - X symbols on is
    looked up by position among the listed suffixes and those the direct comparisons placed among long repeats) -- O(log longest repeat) rounds, the bound the
    reference gets from lcpnext / largelcpthres (/root/reference/src/bwtb3m.cpp:55).  B3M_SPARSE_TIES=0: no list, every shard extends keys to the end (round 3's form:
    hundreds of rounds, same bytes); =300: a list too small for this text (shards that do not fit finish themselves, the rest is finished by key extension)."""
    torch = gpu
    from bwtb3m_amd import engine
    monkeypatch.setenv("B3M_FORCE_SPILL", "4")
    if cap: monkeypatch.setenv("B3M_SPARSE_TIES", cap)
    b = _long_repeat_bases(oracle, 1_600_001, 21)
    buf = oracle.pac_encode(b)
    s = oracle.symbols_from_file_bytes(buf, itype)
    ref = oracle.bwt(s)
    sa = oracle.sa_circular(s)
    n = len(s)
    plan = engine.Plan(n, itype, isasamplingrate=512)
    assert plan.host_tier == 4
    d = torch.from_numpy(np.concatenate([buf, np.zeros(16, dtype=np.uint8)])).cuda()
    out = plan.run(d)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref)
    r = plan.result()
    assert r["tie_elements"] > 80_000, r
    if cap == "": assert r["tie_rounds"] < 40, r
    if cap == "0": assert r["tie_rounds"] > 100, r
    isa = np.empty_like(sa); isa[sa] = np.arange(n)
    def check_side_results():
        rr = plan.result()
        assert rr["p0"] == int(isa[0])
        smp = plan.samples()
        assert len(smp) == (n + 511) // 512 and np.array_equal(isa[smp[:, 1].astype(np.int64)], smp[:, 0].astype(np.int64))
    check_side_results()
    h_text = torch.from_numpy(np.concatenate([buf, np.zeros(64, dtype=np.uint8)])).pin_memory()
    h_out = torch.zeros(n, dtype=torch.uint8).pin_memory()
    plan.run_host(h_text, h_out)                               # the slices have left for host memory when the list is finished: its symbols follow them
    assert np.array_equal(h_out.numpy(), ref)
    check_side_results()
    plan.close()


@pytest.mark.parametrize("itype,ns", [("pacterm", 3), ("pac", 5), ("pacterm", 8)])
def test_shards_finish_long_repeats_from_their_lists(gpu, oracle, itype, ns):
    """The collective tie finish of suffix-range shards WITHOUT their suffix arrays (include/b3m.h: B3M_TIES_LIST; what b3m_compute_bwt numgpus=N does): every shard keeps
    a list of what its four rounds left tied (16-byte records: global rank, position) and of what its direct comparisons placed among long repeats; every shard finishes the
    union by prefix doubling on the list and repairs its own slice, p0 and samples.  All shards on this one GPU, against the oracle."""
    torch = gpu
    import fuzzlib
    from bwtb3m_amd import engine
    b = _long_repeat_bases(oracle, 1_600_001, 33)
    buf = oracle.pac_encode(b)
    s = oracle.symbols_from_file_bytes(buf, itype)
    ref = oracle.bwt(s)
    sa = oracle.sa_circular(s)
    n = len(s)
    d = torch.from_numpy(np.concatenate([buf, np.zeros(16, dtype=np.uint8)])).cuda()
    plans, slices, tied = fuzzlib.run_shards_lists(torch, engine, d, n, itype, ns, isarate=512)
    assert tied > 50_000
    assert np.array_equal(torch.cat(slices).cpu().numpy(), ref)
    assert max(p.result()["tie_rounds"] for p in plans) < 40
    isa = np.empty_like(sa); isa[sa] = np.arange(n)
    p0s = [p.result()["p0"] for p in plans if p.result()["has_p0"]]
    assert p0s == [int(isa[0])]
    smp = np.concatenate([p.samples() for p in plans])
    assert len(smp) == (n + 511) // 512 and np.array_equal(isa[smp[:, 1].astype(np.int64)], smp[:, 0].astype(np.int64))
    for p in plans:
        p.close()


def test_host_tier_is_taken_when_no_merge_tree_meets_mem(gpu, tmp_path):
    """40 Mbase with mem=256m: a merge tree needs 7.25 bytes per symbol whatever its blocks (290 MB + its sorts); the plan goes to the host tier instead of
    exceeding the target (workspace inside mem=), its BWT equals the one-block plan's (LF-verified), and bin/bwtb3m mem=256m writes the files of an
    unconstrained run, byte for byte (bwtonly=1 inside the target; bwtonly=0 says that its LF walks exceed it)."""
    torch = gpu
    from bwtb3m_amd import engine, synth
    itype, d_text, n = synth.config_input_device("c3", size=40_000_000, device="cuda")
    mem = 256 << 20
    plan = engine.Plan(n, itype, mem=mem, isasamplingrate=4096)
    S = plan.host_tier
    assert S >= 2 and plan.warning == "" and plan.workspace_bytes <= mem, (S, plan.warning, plan.workspace_bytes)
    h_text = d_text.cpu().pin_memory()
    h_out = torch.zeros(n, dtype=torch.uint8).pin_memory()
    plan.run_host(h_text, h_out)
    one = engine.Plan(n, itype, isasamplingrate=4096)
    ref = one.run(d_text)
    torch.cuda.synchronize()
    assert one.lf_walk(ref, sarate=0)[:2] == (0, 0) and bool(torch.equal(ref.cpu(), h_out))
    assert one.result()["p0"] == plan.result()["p0"] and np.array_equal(one.samples(), plan.samples())
    plan.close(); one.close()
    inp = tmp_path / "t.pac"
    inp.write_bytes(d_text.cpu().numpy()[: (n - 1) // 4 + 2].tobytes())
    for bwtonly in (1, 0):
        outs = {}
        for tag, extra in (("free", []), ("tier", ["mem=256m"])):
            out = str(tmp_path / ("%s%d.bwt" % (tag, bwtonly)))
            r = subprocess.run([os.path.join(ROOT, "bin", "bwtb3m"), "inputtype=pacterm", "bwtonly=%d" % bwtonly, "verbose=1", "outputfilename=" + out] + extra + [str(inp)],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 0, r.stderr.decode()
            if extra:
                assert b"host tier" in r.stderr and b"cannot be met" not in r.stderr, r.stderr.decode()
                assert (b"[W] mem=" in r.stderr and b"LF walks" in r.stderr) == (bwtonly == 0), r.stderr.decode()
            outs[tag] = _files(out[:-4])
        assert set(outs["free"]) == set(outs["tier"])
        for suf in outs["free"]:
            assert outs["free"][suf] == outs["tier"][suf], (bwtonly, suf)


@pytest.mark.parametrize("itype,bwtonly", [("pacterm", 1), ("pacterm", 0), ("pac", 1), ("bytestream", 1)])
def test_cli_checkbwt_verifies_on_the_device(gpu, oracle, tmp_path, itype, bwtonly):
    """the reference's verifier as a program (/root/reference/src/checkbwt.cpp:248-285: `checkbwt [-i type] X.bwt text`): LF chains between the (rank,pos) pairs of
    X.preisa (bwtonly=1) or X.isa (bwtonly=0), over the text, on the device; "[V] gok=1" for the files bin/bwtb3m wrote, gok=0 once the text differs in one
    symbol; like the reference it leaves with status 0 once the walk has run, and with a message and status 1 when a file is missing."""
    if itype == "bytestream":
        t = oracle.synth_bytes(300_001, 77)
        data = t.tobytes()
    else:
        data = oracle.pac_encode(oracle.synth_repeat_bases(400_002, 12)).tobytes()
    text = tmp_path / "t.in"
    text.write_bytes(data)
    out = str(tmp_path / "t.bwt")
    r = subprocess.run([os.path.join(ROOT, "bin", "bwtb3m"), "inputtype=" + itype, "bwtonly=%d" % bwtonly, "isasamplingrate=4096", "outputfilename=" + out, str(text)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    c = subprocess.run([os.path.join(ROOT, "bin", "checkbwt"), "-i", itype, out, str(text)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert c.returncode == 0 and b"[V] gok=1" in c.stderr and c.stdout == b"", c.stderr.decode()
    bad = bytearray(data)
    bad[len(bad) // 3] ^= 0x10                                   # one base (or byte) of the text changed
    wrong = tmp_path / "wrong.in"
    wrong.write_bytes(bytes(bad))
    c = subprocess.run([os.path.join(ROOT, "bin", "checkbwt"), "-i", itype, out, str(wrong)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert c.returncode == 0 and b"[V] gok=0" in c.stderr and b"[E] failure" in c.stderr, c.stderr.decode()
    c = subprocess.run([os.path.join(ROOT, "bin", "checkbwt"), "-i", itype, str(tmp_path / "nope.bwt"), str(text)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert c.returncode == 1 and c.stderr.strip() != b""
    c = subprocess.run([os.path.join(ROOT, "bin", "checkbwt"), "-i", "klingon", out, str(text)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert c.returncode == 1 and b"unknown/unsupported input type" in c.stderr


@pytest.mark.parametrize("itype", ["pacterm", "pac", "bytestream"])
def test_cli_bwtcomputessa_after_bwtonly1(gpu, oracle, tmp_path, itype):
    """the documented pipeline of bwtonly=1 (/root/reference/README.md:63-67): bwtb3m writes the hint file X.preisa, `bwtcomputessa [sasamplingrate=] [isasamplingrate=] X.bwt`
    (/root/reference/src/bwtcomputessa.cpp:24-58) makes X.sa and X.isa from X.bwt and the hints alone -- here by LF walks on the device, no text: the samples must be the
    oracle's, at rates finer than the hint file's; hints of another text are refused."""
    if itype == "bytestream":
        data = oracle.synth_bytes(250_003, 78).tobytes()
        s = np.frombuffer(data, dtype=np.uint8)
    else:
        buf = oracle.pac_encode(oracle.synth_repeat_bases(350_001, 13))
        data = buf.tobytes()
        s = oracle.symbols_from_file_bytes(buf, itype)
    text = tmp_path / "t.in"
    text.write_bytes(data)
    out = str(tmp_path / "t.bwt")
    r = subprocess.run([os.path.join(ROOT, "bin", "bwtb3m"), "inputtype=" + itype, "bwtonly=1", "outputfilename=" + out, str(text)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and not os.path.exists(out[:-4] + ".sa"), r.stderr.decode()
    c = subprocess.run([os.path.join(ROOT, "bin", "bwtcomputessa"), "sasamplingrate=16", "isasamplingrate=8", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert c.returncode == 0, c.stderr.decode()
    sa = oracle.sa_circular(s)
    n = len(s)
    isa = np.empty_like(sa); isa[sa] = np.arange(n)
    fsa = np.fromfile(out[:-4] + ".sa", dtype=np.uint64)
    fisa = np.fromfile(out[:-4] + ".isa", dtype=np.uint64)
    assert int(fsa[0]) == 16 and int(fsa[1]) == (n + 15) // 16 and np.array_equal(fsa[2:].astype(np.int64), sa[::16])
    assert int(fisa[0]) == 8 and int(fisa[1]) == (n + 7) // 8 and np.array_equal(fisa[2:].astype(np.int64), isa[::8])
    # default rates of the program: 32 / 32
    c = subprocess.run([os.path.join(ROOT, "bin", "bwtcomputessa"), out], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert c.returncode == 0 and int(np.fromfile(out[:-4] + ".sa", dtype=np.uint64)[0]) == 32 and int(np.fromfile(out[:-4] + ".isa", dtype=np.uint64)[0]) == 32
    # hints that belong to another BWT: the chains do not close
    pre = np.fromfile(out[:-4] + ".preisa", dtype=np.uint64).reshape(-1, 2).copy()
    pre[5:40, 0] = pre[5:40, 0][::-1]
    pre.tofile(out[:-4] + ".preisa")
    c = subprocess.run([os.path.join(ROOT, "bin", "bwtcomputessa"), out], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert c.returncode != 0 and b"do not belong together" in c.stderr, c.stderr.decode()


@pytest.mark.gpu
def test_seek_index_consumers_sortpreisa_and_bwtdecodeblock(gpu, oracle, tmp_path):
    """row f4, consumers of the container's seek index: X.preisa -> sorted isa (/root/reference/src/sortPreIsa.cpp:62,95-96), then text[low, low+len)
    out of X.bwt by LF steps from the nearest sample through X.bwt.sparserank (/root/reference/src/bwtdecodeblock.cpp:134-365: findVBlock, the
    per-block symbol counts, inverseSelect, lf, decode; arguments :324-333); the decoder itself compares with the compact text file (:371-386)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(66)
    nb = 150_001
    b = rng.integers(0, 4, nb, dtype=np.uint8)
    b[70_000:71_000] = 0                                         # a homopolymer run: long runs in the BWT, escapes in the container
    pac = tmp_path / "t.pac"
    oracle.pac_encode(b).tofile(pac)
    sym = np.concatenate([b + 1, np.zeros(1, dtype=np.uint8)])   # the text as the index sees it: bases 1..4, terminator 0
    comp = tmp_path / "t.compact"
    oracle.compact_encode(sym, 3).tofile(comp)
    out = tmp_path / "t.bwt"
    r = subprocess.run([os.path.join(root, "bin", "bwtb3m"), "inputtype=pacterm", "bwtonly=1", "isasamplingrate=4096", "outputfilename=%s" % out, str(pac)], capture_output=True)
    assert r.returncode == 0, r.stderr
    sisa = tmp_path / "t.sisa"
    r = subprocess.run([os.path.join(root, "bin", "sortPreIsa"), str(tmp_path / "t.preisa"), str(sisa)], capture_output=True)
    assert r.returncode == 0, r.stderr
    w = np.frombuffer(sisa.read_bytes(), dtype=">u8").reshape(-1, 2)
    assert np.all(np.diff(w[:, 0].astype(np.int64)) > 0)
    sa = oracle.sa_circular(sym)
    assert np.array_equal(sa[w[:, 1].astype(np.int64)], w[:, 0].astype(sa.dtype))          # (pos, rank) pairs
    text = "".join("\nACGT"[int(c)] for c in sym)
    n = len(sym)
    for low, ln in ((0, 100), (5000, 3000), (69_500, 2100), (n - 700, 700), (n - 5000, 4999), (12345, 1)):
        d = subprocess.run([os.path.join(root, "bin", "bwtdecodeblock"), str(out), str(tmp_path / "t.hist"), str(sisa), str(comp), str(low), str(ln)], capture_output=True)
        assert d.returncode == 0, d.stderr[-500:]
        assert d.stdout.decode() == text[low:low + ln], (low, ln)
    assert os.path.exists(str(out) + ".sparserank")
    idx_blocks = len(open(str(out) + ".sparserank", "rb").read()) // (5 * 8)
    assert idx_blocks >= 1
    d = subprocess.run([os.path.join(root, "bin", "bwtdecodeblock"), str(out), str(tmp_path / "t.hist"), str(sisa), str(comp), str(n - 10), "11"], capture_output=True)
    assert d.returncode != 0 and b"low+len=" in d.stderr
