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
