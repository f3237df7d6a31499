"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, option
grammar and defaults mirror the reference CLI, file formats round-trip, errors are loud.  No
compute entry point is exercised successfully here (there is no CPU fallback to exercise)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from bwtb3m_amd import _lib
    if not os.path.exists(_lib.so_path()):
        subprocess.check_call(["make", "-C", ROOT, "-j4", "all"])
    return _lib.lib()


def test_header_symbols_all_exported(L):
    hdr = open(os.path.join(ROOT, "include", "b3m.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(b3m_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 24
    from bwtb3m_amd import _lib
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(L, name), name


def test_environment_switches_live_in_one_place(L):
    # csrc/switches.h: one list, read by host/switches.cpp alone; tools/README.md names every one of them
    names = []
    i = 0
    while True:
        nm = L.b3m_switch_name(i)
        if not nm:
            break
        assert L.b3m_switch_doc(i)
        names.append(nm.decode()); i += 1
    assert len(names) >= 35 and len(set(names)) == len(names) and all(n.startswith("B3M_") for n in names)
    csrc = os.path.join(ROOT, "bwtb3m_amd", "csrc")
    for d, _, files in os.walk(csrc):
        for f in files:
            if f.endswith((".hip", ".h", ".cpp")) and f != "switches.cpp":
                assert "getenv" not in open(os.path.join(d, f)).read(), "%s reads the environment itself" % f
    readme = open(os.path.join(ROOT, "tools", "README.md")).read()
    for n in names:
        assert n in readme, "%s is not listed in tools/README.md" % n


def test_wrong_result_switches_are_not_in_the_product_library():
    # timing-only experiments that skip work (and return wrong results) are compiled into `make experiment` alone: the product
    # library and programs do not even hold the names of their variables
    blobs = [os.path.join(ROOT, "bwtb3m_amd", "libb3m.so")] + [os.path.join(ROOT, "bin", f) for f in sorted(os.listdir(os.path.join(ROOT, "bin")))]
    for b in blobs:
        data = open(b, "rb").read()
        for name in (b"B3M_FINISH_EXPERIMENT", b"B3M_TT_EXPERIMENT", b"B3M_FINISH_NOMAP", b"timing only"):
            assert name not in data, (b, name)
    # (B3M_FINISH_V3=5 was the NOMAP form until round 4: the value no longer selects anything)
    src = open(os.path.join(ROOT, "bwtb3m_amd", "csrc", "k_msd.hip")).read()
    assert "v3 == 5" not in src


def test_no_cpu_fallback(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert not L.b3m_plan_create(1000, 0, 0, 0, 0)
    assert b"no CPU fallback" in L.b3m_last_error()
    h = (ctypes.c_uint64 * 256)()
    assert L.b3m_histogram(None, 10, 0, 0, 10, h, None) != 0


def test_library_missing_is_loud(monkeypatch, tmp_path):
    from bwtb3m_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_SO", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.LibraryMissing):
        _lib.lib()


def test_input_type_parsing(L):
    names = ["bytestream", "compactstream", "pac", "pacterm", "lz4", "utf-8"]   # /root/reference/src/bwtb3m.cpp:43
    for i, nme in enumerate(names):
        assert L.b3m_parse_input_type(nme.encode()) == i
        assert L.b3m_input_type_name(i) == nme.encode()
    assert L.b3m_parse_input_type(b"fastq") == -1
    assert L.b3m_last_error() == b"libmaus2::suffixsort::bwtb3m::BwtMergeSort::computeBwt: unknown/unsupported input type"   # /root/reference/src/checkbwt.cpp:274


def test_options_defaults(L):
    from bwtb3m_amd import _lib
    o = _lib.Options()
    L.b3m_options_init(ctypes.byref(o))
    # /root/reference/README.md:38-49
    assert o.inputtype == b"bytestream" and o.sasamplingrate == 32 and o.isasamplingrate == 262144
    assert o.bwtonly == 0 and o.copyinputtomemory == 0 and o.largelcpthres == 16384 and o.verbose == 0
    assert o.numthreads >= 1


def test_python_options_mirror():
    from bwtb3m_amd import ArgInfo, BwtMergeSortOptions, B3MError
    ai = ArgInfo(["bwtb3m", "inputtype=pacterm", "mem=2g", "bwtonly=1", "outputfilename=x.bwt", "in.pac", "extra"])
    assert ai.restargs == ["in.pac", "extra"] and ai.argmap["mem"] == "2g"
    o = BwtMergeSortOptions(ai)
    assert o.fn == "in.pac" and o.inputtype == "pacterm" and o.mem == 2 << 30 and o.outfn == "x.bwt"
    assert o.isasamplingrate == 64 and o.sasamplingrate == 32       # ChangeLog:281: 64 when bwtonly=1
    o2 = BwtMergeSortOptions(ArgInfo(["bwtb3m", "f"]))
    assert o2.isasamplingrate == 262144 and o2.outfn.endswith(".bwt") and o2.sparsetmpfilenamebase == o2.tmpfilenamebase
    assert re.match(r"bwtb3m_.+_\d+_\d+$", o2.tmpfilenamebase)
    with pytest.raises(B3MError):
        BwtMergeSortOptions(ArgInfo(["bwtb3m", "inputtype=fastq", "f"]))
    with pytest.raises(B3MError):
        BwtMergeSortOptions(ArgInfo(["bwtb3m", "mem=12x", "f"]))
    assert ArgInfo(["p", "-h"]).helpRequested()


def _run(args, **kw):
    return subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


def test_cli_help_and_errors(L):
    exe = os.path.join(ROOT, "bin", "bwtb3m")
    r = _run([exe])
    # no positional: usage thrown as an exception -> stderr, EXIT_FAILURE; stdout unused
    assert r.returncode == 1 and r.stdout == b""
    txt = r.stderr.decode()
    for opt in ["inputtype=[<bytestream>] (bytestream,compactstream,pac,pacterm,lz4,utf-8)", "outputfilename=", "sasamplingrate=[32]",
                "isasamplingrate=[262144]", "mem=", "numthreads=", "bwtonly=[0]", "tmpprefix=", "sparsetmpprefix=[tmpprefix]",
                "copyinputtomemory=[0]", "largelcpthres=[16384]", "verbose=[0]"]:
        assert opt in txt, opt
    assert "usage: " in txt and "[options] <inputfile>" in txt
    assert _run([exe, "-h"]).returncode == 1
    r = _run([exe, "inputtype=nosuch", "/dev/null"])
    assert r.returncode == 1 and b"unknown/unsupported input type" in r.stderr
    r = _run([exe, "/nonexistent/file"])
    assert r.returncode == 1 and b"cannot open" in r.stderr


def test_rl_container_round_trip(L, tmp_path):
    rng = np.random.default_rng(5)
    cases = [np.zeros(1, np.uint8), np.array([7], np.uint8), rng.integers(0, 256, 100000).astype(np.uint8),
             rng.integers(1, 5, 300000).astype(np.uint8), np.repeat(rng.integers(0, 4, 50).astype(np.uint8), 9000)]
    for i, a in enumerate(cases):
        fn = str(tmp_path / ("t%d.bwt" % i)).encode()
        assert L.b3m_rl_write(fn, a.ctypes.data, len(a), 4096) == 0, L.b3m_last_error()
        n = ctypes.c_uint64()
        assert L.b3m_rl_length(fn, ctypes.byref(n)) == 0 and n.value == len(a)
        out = np.zeros(len(a), np.uint8)
        assert L.b3m_rl_read(fn, out.ctypes.data, len(a), ctypes.byref(n)) == 0, L.b3m_last_error()
        assert np.array_equal(out, a)
        # first 8 bytes: n, big-endian; last 8 bytes: index offset
        raw = open(fn, "rb").read()
        assert int.from_bytes(raw[:8], "big") == len(a)
        assert 8 <= int.from_bytes(raw[-8:], "big") < len(raw)
        # the named consumer decodes it (bwtb3mdecoderl, bytestream mode)
        r = _run([os.path.join(ROOT, "bin", "bwtb3mdecoderl"), fn.decode()])
        assert r.returncode == 0 and r.stdout == a.tobytes()
    assert L.b3m_rl_length(b"/nonexistent", ctypes.byref(n)) != 0
    bad = tmp_path / "bad.bwt"
    bad.write_bytes(b"\x00" * 10)
    assert L.b3m_rl_read(str(bad).encode(), None, 0, ctypes.byref(n)) != 0
    r = _run([os.path.join(ROOT, "bin", "bwtb3mdecoderl")])
    assert r.returncode == 1 and b"usage:" in r.stderr
    r = _run([os.path.join(ROOT, "bin", "bwtb3mdecoderl"), "inputtype=pac", str(bad)])
    assert r.returncode == 1 and b"currently not supported for bwtb3mdecode" in r.stderr


def test_synth_generators_agree(oracle):
    from bwtb3m_amd import synth
    for nb in [1, 3, 4, 5, 31, 32, 33, 1000, 100003]:
        a = synth.pac_file_bytes(nb, synth.SEED_BASE + 1)
        assert np.array_equal(a, oracle.pac_encode(oracle.synth_bases(nb, 0xB3B3B3B3 + 1)))
        assert np.array_equal(a, synth.pac_file_bytes_torch(nb, synth.SEED_BASE + 1, device="cpu", chunk_words=777).numpy())
    it, buf, n = synth.config_input("c2")
    assert it == "pacterm" and len(buf) == 1_250_002 and n == 5_000_001
    it, buf, n = synth.config_input("c1")
    assert it == "bytestream" and n == 1 << 20 and np.array_equal(buf, oracle.synth_bytes(1 << 20, 0xB3B3B3B3))


def test_bwtb3mtobwa_conversion(L, oracle, tmp_path):
    """X.bwt (+X.sa) of a pacterm construction -> BWA's .bwt/.sa layout.  The inputs are
    produced from the oracle here (no GPU needed); tests/test_gpu_boundary.py runs the real pipeline."""
    b = oracle.synth_bases(5003, 19)
    s = np.concatenate([b + 1, [0]]).astype(np.uint8)
    sa = oracle.sa_circular(s)
    bw = oracle.bwt(s)
    n = len(s)
    fn = str(tmp_path / "x.pac.bwt")
    assert L.b3m_rl_write(fn.encode(), bw.ctypes.data, n, 4096) == 0
    rate = 32
    smp = sa[::rate].astype(np.uint64)
    np.concatenate([np.array([rate, len(smp)], dtype=np.uint64), smp]).tofile(str(tmp_path / "x.pac.sa"))
    ob, os_ = str(tmp_path / "o.bwt"), str(tmp_path / "o.sa")
    r = _run([os.path.join(ROOT, "bin", "bwtb3mtobwa"), fn, ob, os_])
    assert r.returncode == 0, r.stderr.decode()
    raw = np.fromfile(ob, dtype=np.uint8)
    head = raw[:40].view(np.uint64)
    primary = int(np.nonzero(bw == 0)[0][0])
    assert head[0] == primary == int(np.nonzero(sa == 0)[0][0])
    counts = np.bincount(b, minlength=4)
    assert head[1:5].tolist() == np.cumsum(counts).tolist()           # L2[1..4]
    words = raw[40:].view(np.uint32)
    sym = ((words[:, None] >> np.arange(30, -2, -2, dtype=np.uint32)[None, :]) & 3).astype(np.uint8).reshape(-1)[: n - 1]
    assert np.array_equal(sym, np.delete(bw, primary) - 1)           # BWT without the terminator entry, first symbol in the MSBs
    fsa = np.fromfile(os_, dtype=np.uint64)
    assert fsa[0] == primary and fsa[1:5].tolist() == np.cumsum(counts).tolist() and fsa[5] == rate and fsa[6] == n - 1
    n_sa = (n - 1 + rate) // rate
    assert np.array_equal(fsa[7:].astype(np.int64), sa[rate::rate][: n_sa - 1])
    # errors: usage, missing .sa, wrong alphabet
    assert _run([os.path.join(ROOT, "bin", "bwtb3mtobwa"), fn]).returncode == 1
    os.remove(str(tmp_path / "x.pac.sa"))
    r = _run([os.path.join(ROOT, "bin", "bwtb3mtobwa"), fn, ob, os_])
    assert r.returncode == 1 and b"cannot open" in r.stderr
    bad = np.full(10, 7, dtype=np.uint8)
    fb = str(tmp_path / "bad.bwt")
    assert L.b3m_rl_write(fb.encode(), bad.ctypes.data, 10, 4096) == 0
    r = _run([os.path.join(ROOT, "bin", "bwtb3mtobwa"), fb, ob, os_])
    assert r.returncode == 1 and b"not the BWT of a pacterm text" in r.stderr


def test_bench_roofline_helper():
    # the `roofline` object of the bench line: dominant family, achieved = algorithmic bytes / device time
    from bwtb3m_amd import benchutil
    fam = {}
    kt = [{"name": "msd_scatter", "ms": 16.0, "launches": 1, "alg_bytes": 48_000_000_016},
          {"name": "bucket_finish", "ms": 12.0, "launches": 1, "alg_bytes": 27_000_000_009},
          {"name": "scan", "ms": 0.3, "launches": 6, "alg_bytes": 1000}]
    benchutil.add_kernel_times(fam, kt); benchutil.add_kernel_times(fam, kt)
    r = benchutil.dominant_roofline(fam, 2, 0.104, 3_000_000_000, 0, 0, "/nonexistent")
    assert r["family"] == "msd_scatter" and r["kernel"] == "b3m::k_l1_gather<1024, 4>" and r["bound"] == "hbm"
    assert abs(r["achieved"] - 3000.0) < 1.0 and abs(r["frac"] - 0.375) < 1e-3 and r["traffic"] is None
    assert r["launches_per_step"] == 1 and r["alg_bytes_per_launch"] == 48_000_000_016
    assert abs(r["job_frac"] - 28.5e9 / 0.052 / 1e9 / 8000.0) < 1e-4
    assert set(r["families_ms_per_step"]) == {"msd_scatter", "bucket_finish", "scan"}
    # the committed PMC passes name the kernels by their full instantiation: every family that can be the dominant one of the
    # one-block C3 step must be found EXACTLY in the newest file (round 4's line had `traffic: null` after a template parameter was added)
    import glob, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    newest = sorted(glob.glob(os.path.join(root, "profiles", "r0*_pmc_traffic.json")))[-1]
    tj = json.load(open(newest))
    for f in ("msd_scatter", "msd_scatter_text", "bucket_finish"):
        assert benchutil.KERNEL_OF[f] in tj["kernels"], (f, benchutil.KERNEL_OF[f], newest)
    fam2 = {"bucket_finish": {"ms": 11.0, "launches": 1, "alg_bytes": 27_000_000_009}, "scan": {"ms": 0.3, "launches": 6, "alg_bytes": 1000}}
    r2 = benchutil.dominant_roofline(fam2, 1, 0.030, 3_000_000_000, 0, tj["workload_n"], root)
    assert r2["traffic"] is not None and r2["traffic"] > 20e9
    # a kernel whose template arguments changed since the passes were taken still resolves (same template, most bytes)
    assert benchutil.pmc_kernel_name("b3m::k_finish_v3<3, 512, false, 7>", tj["kernels"]) == benchutil.KERNEL_OF["bucket_finish"]
    assert benchutil.pmc_kernel_name("b3m::k_no_such_kernel<1>", tj["kernels"]) is None


def test_bench_launcher_counts_gpus_without_touching_hip(monkeypatch):
    """the self-launching parent of `bench.py --gpus N` must not initialise the GPU; it counts devices from the KFD topology in sysfs (0 where that is
    absent, as in this container: the launcher then lets the ranks decide) and honours HIP_VISIBLE_DEVICES"""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    n = bench.benchutil_count_gpus()
    assert isinstance(n, int) and n >= 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    assert bench.benchutil_count_gpus() <= 1
    src = open(os.path.join(root, "bench.py")).read()
    launcher = src[src.index("def launch_ranks"):src.index("def main")]
    code = "\n".join(l.split("#")[0] for l in launcher.split('"""')[2].splitlines())      # the body behind the docstring, comments cut
    assert "import torch" not in code and "torch.cuda" not in code, "the launcher must not import torch or touch torch.cuda"


def test_genome_like_recipes_on_the_cpu():
    """both recipes of the repeat-rich bench legs (synth.genome_like_device): deterministic, the right size, and the harsher one plants more"""
    import torch
    from bwtb3m_amd import synth
    nb = 2_000_000
    a = synth.genome_like_device(nb, 7, "cpu")
    b = synth.genome_like_device(nb, 7, "cpu")
    h = synth.genome_like_device(nb, 7, "cpu", repeats=0.5)
    assert a.numel() == nb // 4 + 1 + 64 + (1 if nb % 4 == 0 else 0) and h.numel() == a.numel()
    # (two draws with one seed agree except where planted copies overlap: which copy's base wins there is the scatter's choice -- the listed-suffix counts of two
    # bench runs differ in the sixth digit for that reason; every run is LF-verified against the text it got)
    assert int((a != b).sum()) < a.numel() // 20
    assert int((a != h).sum()) > a.numel() // 10
