"""Fixed-seed slices of the randomised sweeps (tools/fuzz*.py run the same cases, tests/fuzzlib.py, for hours): in round 2 the sweeps found
three wrong results that no fixed-size test had reached, so a slice of each now runs with every `pytest -m gpu`.  The alternative code
paths of the library (forced digit widths, scattering level 0, generic finish, one-step tie fix) are drawn by the cases themselves."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(cases):
    bad = [d for ok, d in cases if not ok]
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("seed", [101, 102, 103])
def test_fuzz_small_cases_against_the_oracle(gpu, oracle, seed):
    from bwtb3m_amd import engine
    import fuzzlib
    rng = np.random.default_rng(seed)
    _run([fuzzlib.case_small(rng, oracle, engine) for _ in range(10)])


@pytest.mark.parametrize("seed", [201, 202])
def test_fuzz_suffix_range_shards_against_the_oracle(gpu, oracle, seed):
    from bwtb3m_amd import engine
    import fuzzlib
    rng = np.random.default_rng(seed)
    _run([fuzzlib.case_shards(rng, oracle, engine, gpu) for _ in range(8)])


@pytest.mark.parametrize("seed", [301, 302])
def test_fuzz_large_cases_lf_verified(gpu, seed, monkeypatch):
    """24 - 250 Mbase: one block (LF walk from its own samples) against a merge tree / mem=-chosen blocks / shards / host-to-host runs.
    B3M_DEBUG=2: every stage of the block sort is drained and named on stderr, so a GPU fault here names its stage."""
    from bwtb3m_amd import engine
    import fuzzlib
    monkeypatch.setenv("B3M_DEBUG", "2")
    rng = np.random.default_rng(seed)
    _run([fuzzlib.case_large(rng, engine, gpu, 24_000_000, 250_000_000) for _ in range(5)])


def test_fuzz_byte_alphabets_large(gpu):
    from bwtb3m_amd import engine
    import fuzzlib
    rng = np.random.default_rng(401)
    _run([fuzzlib.case_bytes_large(rng, engine, gpu, 5_000_000, 40_000_000) for _ in range(4)])
