"""N>1 path of the multi-GPU driver under gloo on the CPU (world sizes 2, 3, 4): tree plan, chain
ranges, gap-bucket all-to-all, BWT(R) range all-to-all, slice bookkeeping, gt bits, p0.  The compute
steps are the oracle's (tests/oracle_backend.py); what is tested is bwtb3m_amd/distributed.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, itype, nb, seed, repeat, q, leaves=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from oracle_backend import OracleBackend
        from bwtb3m_amd import distributed as D
        b = O.synth_repeat_bases(nb, seed) if repeat else O.synth_bases(nb, seed)
        if itype == "bytestream":
            # a byte text with structure: bases mapped to letters, a 400-symbol repeat, a few rare symbols (0 and 255 among them)
            buf = np.frombuffer(b"ACGT", dtype=np.uint8)[b].copy()
            buf[nb // 2: nb // 2 + 400] = buf[100:500]
            buf[::97] = 255; buf[5::211] = 0; buf[7::53] = 10
        else:
            buf = O.pac_encode(b)
        sym = O.symbols_from_file_bytes(buf, itype)
        n = len(sym)
        job = D.DistributedBwt(n, itype, OracleBackend(sym, itype), torch.device("cpu"), leaves=leaves)
        out, off, p0 = job.run(torch.from_numpy(buf))
        full = D.gather_result(out).numpy()
        ref = O.bwt(sym)
        sa = O.sa_circular(sym)
        ok = np.array_equal(full, ref)
        # my slice sits at its offset
        ok = ok and np.array_equal(ref[off: off + out.numel()], out.numpy())
        if itype == "pacterm":
            ok = ok and p0 == int(np.nonzero(sa == 0)[0][0])
        q.put((rank, bool(ok), int(out.numel())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("itype,nb,repeat", [("pacterm", 5000, False), ("pac", 4097, False), ("pacterm", 6000, True), ("bytestream", 4001, False)])
def test_distributed_driver_gloo(world, itype, nb, repeat, leaves=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, itype, nb, 17 + world, repeat, q, leaves)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    for p in procs:
        p.join(timeout=240)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail("worker hung")
        assert p.exitcode == 0
    while not q.empty():
        res.append(q.get())
    assert len(res) == world and all(ok for _, ok, _ in res), res
    n = nb + (1 if itype == "pacterm" else 0)
    assert sum(sz for _, _, sz in res) == n


def test_distributed_driver_gloo_eight_ranks():
    """the world size of the reference's configs[3] (8 GPUs): three merge levels, groups of 2, 4 and 8 ranks, one leaf per rank"""
    test_distributed_driver_gloo(8, "pacterm", 6001, False)
    test_distributed_driver_gloo(8, "bytestream", 5003, False)


@pytest.mark.parametrize("world,leaves", [(1, 4), (2, 4), (2, 5), (3, 7), (4, 8)])
def test_distributed_driver_gloo_more_leaves_than_ranks(world, leaves):
    """"GPU g sorts blocks g, g + P, ...": B >= P leaf blocks, a rank holds slices of both sides of a merge, groups of one rank (world 1:
    every collective of a merge still runs, on a group of one); gt bits travel holder -> walkers' chain ranges -> the parent's holder"""
    test_distributed_driver_gloo(world, "pacterm", 5003, False, leaves)
    test_distributed_driver_gloo(world, "bytestream", 3001, False, leaves)
    if world == 2:
        test_distributed_driver_gloo(world, "pacterm", 6000, True, leaves)
        test_distributed_driver_gloo(world, "pac", 4097, False, leaves)


def test_tree_plan_shapes():
    sys.path.insert(0, ROOT)
    from bwtb3m_amd.distributed import TreePlan, split_even, chains_for
    p = TreePlan(3_000_000_001, 8)
    assert len(p.leaves) == 8 and len(p.levels) == 3 and [len(l) for l in p.levels] == [4, 2, 1]
    assert sum(l.m for l in p.leaves) == 3_000_000_001
    top = p.levels[-1][0]
    assert top.group == list(range(8)) and top.L.m + top.R.m == 3_000_000_001
    # chains tile R exactly
    for mg in p.merges:
        assert (mg.num_chains - 1) * mg.chain_len < mg.R.m <= mg.num_chains * mg.chain_len
    p3 = TreePlan(1000, 3)          # odd: the last leaf is carried one level
    assert [len(l) for l in p3.levels] == [1, 1] and p3.levels[1][0].group == [0, 1, 2]
    # more leaves than ranks: round robin, groups = the ranks that own a leaf of either side, the holder of a node = the owner of its first leaf
    p24 = TreePlan(1000, 2, leaves=4)
    assert [(mg.L.leaves, mg.R.leaves, mg.group, mg.L.holder, mg.R.holder) for mg in p24.merges] == [([0], [1], [0, 1], 0, 1), ([2], [3], [0, 1], 0, 1), ([0, 1], [2, 3], [0, 1], 0, 0)]
    assert p24.leaves_of(1) == [1, 3] and p24.root.key == 0
    p14 = TreePlan(1000, 1, leaves=4)
    assert all(mg.group == [0] for mg in p14.merges) and len(p14.merges) == 3
    assert split_even(10, 3) == [0, 3, 6, 10]
    assert chains_for(100)[0] == 32
