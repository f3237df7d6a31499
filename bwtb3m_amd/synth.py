"""Deterministic synthetic inputs of the BASELINE configs: splitmix64 started at
seed 0xB3B3B3B3 + config index, uniform i.i.d.  32 bases per 64-bit output, most significant pair
first, which makes the BWA .pac payload the big-endian bytes of the PRNG stream."""
import numpy as np

SEED_BASE = 0xB3B3B3B3

CONFIGS = {
    # name: (inputtype, number of bases/bytes, config index)
    "c1": ("bytestream", 1 << 20, 0),
    "c2": ("pacterm", 5_000_000, 1),
    "c3": ("pacterm", 3_000_000_000, 2),
    "c5": ("pacterm", 30_000_000_000, 4),      # BASELINE configs[4]: n = 30,000,000,001 > 2^32 (36-bit record positions, suffix-range shards)
}


def splitmix64(seed, count, start=0):
    idx = np.arange(start + 1, start + count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & ((1 << 64) - 1)) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def pac_file_bytes(nbases, seed, chunk_words=1 << 24):
    """BWA .pac image of `nbases` uniform bases: ceil(l/4) data bytes,
    an extra 0 byte if l%4==0, then the byte l%4."""
    nwords = (nbases + 31) // 32
    ndata = (nbases + 3) // 4
    out = np.zeros(nbases // 4 + 2, dtype=np.uint8)
    pos = 0
    for w0 in range(0, nwords, chunk_words):
        cnt = min(chunk_words, nwords - w0)
        b = splitmix64(seed, cnt, start=w0).astype(">u8").view(np.uint8)
        take = min(len(b), ndata - pos)
        out[pos:pos + take] = b[:take]
        pos += take
    if nbases % 4:
        # clear the unused low bit pairs of the last data byte
        keep = nbases % 4
        out[ndata - 1] &= np.uint8((0xFF << (2 * (4 - keep))) & 0xFF)
    out[-1] = nbases % 4
    return out


def byte_file_bytes(nbytes, seed):
    return splitmix64(seed, (nbytes + 7) // 8).astype(">u8").view(np.uint8)[:nbytes].copy()


def config_input(name, size=None):
    """(inputtype, file bytes, number of symbols) of a BASELINE config; `size` overrides the length"""
    itype, n, idx = CONFIGS[name]
    if size:
        n = int(size)
    seed = SEED_BASE + idx
    if itype == "bytestream":
        return itype, byte_file_bytes(n, seed), n
    return itype, pac_file_bytes(n, seed), n + (1 if itype == "pacterm" else 0)


def _splitmix64_torch(seed, count, start, device):
    """same stream as splitmix64() in int64 two's-complement arithmetic (torch has no uint64 math)"""
    import torch

    def s64(x):
        x &= (1 << 64) - 1
        return x - (1 << 64) if x >= (1 << 63) else x

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)

    idx = torch.arange(start + 1, start + count + 1, dtype=torch.int64, device=device)
    z = idx * s64(0x9E3779B97F4A7C15) + s64(seed)
    z = (z ^ lsr(z, 30)) * s64(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * s64(0x94D049BB133111EB)
    return z ^ lsr(z, 31)


def pac_file_bytes_torch(nbases, seed, device="cuda", chunk_words=1 << 26):
    """pac_file_bytes() computed with torch on `device`; returns a uint8 tensor (same bytes)"""
    import torch
    nwords = (nbases + 31) // 32
    ndata = (nbases + 3) // 4
    out = torch.zeros(nbases // 4 + 2, dtype=torch.uint8, device=device)
    pos = 0
    for w0 in range(0, nwords, chunk_words):
        cnt = min(chunk_words, nwords - w0)
        z = _splitmix64_torch(seed, cnt, w0, device)
        b = z.view(torch.uint8).view(-1, 8).flip(1).reshape(-1)     # big-endian bytes
        take = min(b.numel(), ndata - pos)
        out[pos:pos + take] = b[:take]
        pos += take
    if nbases % 4:
        keep = nbases % 4
        out[ndata - 1] &= (0xFF << (2 * (4 - keep))) & 0xFF
    out[-1] = nbases % 4
    return out


def config_input_device(name, size=None, device="cuda"):
    """like config_input() but the file image is generated on the device (torch tensor)"""
    import torch
    itype, n, idx = CONFIGS[name]
    if size:
        n = int(size)
    seed = SEED_BASE + idx
    if itype == "bytestream":
        z = _splitmix64_torch(seed, (n + 7) // 8, 0, device)
        return itype, z.view(torch.uint8).view(-1, 8).flip(1).reshape(-1)[:n].contiguous(), n
    return itype, pac_file_bytes_torch(n, seed, device), n + (1 if itype == "pacterm" else 0)


def genome_like_device(nb, seed=7, device="cuda", repeats=0.15):
    """nb bases with the repeat structure of a genome, on the device (the recipe of tools/repeat_test.py, vectorised): uniform background;
    a 300-bp element at 10 % divergence over ~10 % of the text; a 6-kb element at 3 % over ~5 %; six 200-kb segmental duplications at
    0.1 %; microsatellites (unit 1-6, 20-200 copies) every 100 kb; one satellite array of 171 bp x 3000 at 1 %.  Returns the .pac image
    (uint8 tensor, padded) for inputtype pac / pacterm.  NOT the timed metric (BASELINE: uniform); reported beside it.
    repeats=0.5: the harsher recipe, nearer to a human genome (about half of it interspersed repeats): the 300-bp element at 12 % over 10 % of the text (Alu-like) and an
    older family of it at 16 % over 8 %, a 6-kb element at 8 % over 17 % (L1-like; truncated copies: a random stretch of 0.5-6 kb of it), a 2-kb element at 12 % over
    8 %, then the duplications, microsatellites and the satellite array as above: ~45 % of the positions planted."""
    import torch
    g = torch.Generator(device=device); g.manual_seed(seed)
    b = torch.empty(nb, dtype=torch.uint8, device=device)
    for o in range(0, nb, 1 << 28):
        m_ = min(1 << 28, nb - o)
        b[o:o + m_] = torch.randint(0, 4, (m_,), device=device, dtype=torch.uint8, generator=g)

    def plant(unit, copies, rate, truncated=False):
        ln = unit.numel()
        per = max(1, min(1 << 16, (1 << 27) // ln)) if repeats >= 0.3 else 1 << 16       # (copies at a time; the default recipe keeps round 4's sequence of draws: the same text)
        for c0 in range(0, copies, per):
            c = min(per, copies - c0)
            p = torch.randint(0, nb - ln, (c,), device=device, generator=g)
            seg = unit.repeat(c, 1)
            mut = torch.rand((c, ln), device=device, generator=g) < rate
            seg = torch.where(mut, (seg + torch.randint(1, 4, (c, ln), device=device, dtype=torch.uint8, generator=g)) & 3, seg)
            at = p[:, None] + torch.arange(ln, device=device)[None, :]
            if truncated:               # a copy is a random stretch of the element (its 3' part, as L1 copies are): columns [first, ln)
                first = torch.randint(0, ln - ln // 12, (c,), device=device, generator=g)
                keep = torch.arange(ln, device=device)[None, :] >= first[:, None]
                b[at[keep]] = seg[keep]
            else:
                b[at.reshape(-1)] = seg.reshape(-1)

    if repeats >= 0.3:
        alu = torch.randint(0, 4, (300,), device=device, dtype=torch.uint8, generator=g)
        plant(alu, nb // 3750, 0.16)                                      # the older family first: younger copies land on top of it
        plant(torch.randint(0, 4, (6000,), device=device, dtype=torch.uint8, generator=g), int(nb * 0.17 / 3250), 0.08, truncated=True)
        plant(torch.randint(0, 4, (2000,), device=device, dtype=torch.uint8, generator=g), nb // 25000, 0.12)
        plant(alu, nb // 3000, 0.12)
    else:
        plant(torch.randint(0, 4, (300,), device=device, dtype=torch.uint8, generator=g), nb // 3000, 0.10)
        plant(torch.randint(0, 4, (6000,), device=device, dtype=torch.uint8, generator=g), nb // 120000, 0.03)
    if nb > 400000:
        plant(torch.randint(0, 4, (200000,), device=device, dtype=torch.uint8, generator=g), 6, 0.001)
    nms = nb // 100000                                           # microsatellites: unit u, c copies, as a masked (nms, 1200) block
    if nms:
        u = torch.randint(1, 7, (nms,), device=device, generator=g)
        c = torch.randint(20, 200, (nms,), device=device, generator=g)
        p = torch.randint(0, nb - 1200, (nms,), device=device, generator=g)
        unit = torch.randint(0, 4, (nms, 6), device=device, dtype=torch.uint8, generator=g)
        j = torch.arange(1200, device=device)[None, :]
        val = torch.gather(unit, 1, j % u[:, None])
        mask = j < (u * c)[:, None]
        idx = (p[:, None] + j)[mask]
        b[idx] = val[mask]
    if nb > 171 * 3000:
        sat = torch.randint(0, 4, (171,), device=device, dtype=torch.uint8, generator=g).repeat(3000)
        mut = torch.rand(sat.numel(), device=device, generator=g) < 0.01
        sat = torch.where(mut, (sat + torch.randint(1, 4, (sat.numel(),), device=device, dtype=torch.uint8, generator=g)) & 3, sat)
        p = int(torch.randint(0, nb - sat.numel(), (1,), device=device, generator=g))
        b[p:p + sat.numel()] = sat
    pad = (-nb) % 4
    bb = torch.cat([b, torch.zeros(pad, dtype=torch.uint8, device=device)]).view(-1, 4)
    pac = ((bb[:, 0] << 6) | (bb[:, 1] << 4) | (bb[:, 2] << 2) | bb[:, 3]).to(torch.uint8)
    tail = torch.tensor([0] * (1 if nb % 4 == 0 else 0) + [nb % 4], dtype=torch.uint8, device=device)
    return torch.cat([pac, tail, torch.zeros(64, dtype=torch.uint8, device=device)])
