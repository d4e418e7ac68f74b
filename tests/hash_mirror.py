"""numpy mirror of the attention-dropout keep decisions (wav2vec-s_amd/csrc/attn_common.h: pair_hash_pm / seed_mix; word index
and field selection as in attention2.hip).  Test infrastructure: tests/test_host_cpu.py checks its statistics (within one
seed and ACROSS seeds), tests/test_kernels_gpu.py checks that it equals the kernels' own decisions."""
import numpy as np

HASH_K = 0x9E3779B1
M32 = 0xFFFFFFFF


def seed_mix(s1: int) -> int:
    m = ((s1 ^ (s1 >> 15)) * 0x85EBCA77) & M32
    return m ^ (m >> 13)


def hash_words(seed: int, idx: np.ndarray) -> np.ndarray:
    """pair_hash_pm for an array of 32-bit word indices."""
    s0, s1 = seed & M32, (seed >> 32) & M32
    x0 = (idx.astype(np.uint64) * HASH_K + s0) & M32
    x = x0 ^ (x0 >> np.uint64(16)) ^ np.uint64(seed_mix(s1))
    h = ((x & np.uint64(0xFFFFFF)) * np.uint64(0xB5352D) + x0) & np.uint64(M32)
    return (h ^ (h >> np.uint64(15))).astype(np.uint64)


def thr16(p: float) -> int:
    return int(float(np.float32(p)) * 4294967296.0) >> 16


def attn_keep(seed: int, B: int, H: int, N: int, p: float, Ns=None) -> np.ndarray:
    """keep[b, h, q, key] (bool) for q in [0, Ns): word ((b*H + h)*Ns + q) * ceil(N/2) + key // 2, low half for even keys."""
    Ns = N if Ns is None else Ns
    Nh = (N + 1) // 2
    rows = (np.arange(B * H * Ns, dtype=np.uint64) * Nh)[:, None] + np.arange(Nh, dtype=np.uint64)[None, :]
    h = hash_words(seed, rows)
    lo, hi = (h & np.uint64(0xFFFF)) >= thr16(p), (h >> np.uint64(16)) >= thr16(p)
    keep = np.stack([lo, hi], axis=-1).reshape(B * H * Ns, 2 * Nh)[:, :N]
    return keep.reshape(B, H, Ns, N)


_KR = [(i & 3) + 8 * (i >> 2) + 4 * w_ for i in range(16) for w_ in range(2)]     # key row of dword 2i + w (attn_common.h, AttnP::drop_bits)


def decode_drop_bits(bits, B, H, N):
    """The keep-mask records a forward launch parked in w2vs_attn_desc.drop_bits -> bool [B, H, N, N] (torch, on the records'
    device).  Blocks no query can see are never written: fill the buffer with -1 ('keep') before the launch."""
    import torch
    nT = (N + 31) // 32
    w = bits.view(B * H, nT, nT, 32).long() & 0xFFFFFFFF
    qb = (w.unsqueeze(-1) >> torch.arange(32, device=bits.device)) & 1             # [BH, qt, kt, dword, query bit]
    inv = torch.empty(32, dtype=torch.long)
    inv[torch.tensor(_KR)] = torch.arange(32)
    qb = qb.index_select(3, inv.to(bits.device))                                   # [BH, qt, kt, key row, query]
    keep = qb.permute(0, 1, 4, 2, 3).reshape(B * H, nT * 32, nT * 32)[:, :N, :N]
    return keep.reshape(B, H, N, N).bool()
