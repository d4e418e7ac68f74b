"""Host-side integer / RNG half of the forward pass.

The reference draws these on the host from three different generators, and "bit-exact mask
indices / negative sampling" means reproducing the draws *and their order*:
  numpy global RandomState : compute_mask_indices, then one random() per encoder layer (LayerDrop)
  python ``random``         : the two randint() of context sampling
  torch CPU generator      : the single randint of sample_negatives
The draws stay on the host here as well; only their results travel to the GPU.
"""
import random
from typing import List, Optional, Tuple

import numpy as np
import torch


def compute_mask_indices(shape: Tuple[int, int], padding_mask: Optional[torch.Tensor], mask_prob: float,
                         mask_length: int, mask_type: str = "static", mask_other: float = 0.0, min_masks: int = 0,
                         no_overlap: bool = False, min_space: int = 0) -> np.ndarray:
    """Same contract as fairseq.data.data_utils.compute_mask_indices
    (fs/data/data_utils.py:389-513): identical results for identical numpy global-RNG state.
    ``no_overlap=True`` is rejected: that branch of the reference calls the removed ``np.int``
    (data_utils.py:481) and cannot run on current numpy either."""
    if no_overlap:
        raise NotImplementedError("no_overlap masks: the reference branch (data_utils.py:469-488) depends on np.int")
    bsz, all_sz = shape
    rng = np.random  # the GLOBAL RandomState, as in the reference
    all_num_mask = max(min_masks, int(mask_prob * all_sz / float(mask_length) + rng.rand()))
    rows: List[np.ndarray] = []
    for i in range(bsz):
        if padding_mask is not None:
            sz = all_sz - int(padding_mask[i].long().sum().item())
            num_mask = max(min_masks, int(mask_prob * sz / float(mask_length) + rng.rand()))
        else:
            sz, num_mask = all_sz, all_num_mask
        if mask_type == "static":
            lengths = np.full(num_mask, mask_length)
        elif mask_type == "uniform":
            lengths = rng.randint(mask_other, mask_length * 2 + 1, size=num_mask)
        elif mask_type == "normal":
            lengths = np.asarray([max(1, int(round(x))) for x in rng.normal(mask_length, mask_other, size=num_mask)])
        elif mask_type == "poisson":
            lengths = np.asarray([int(round(x)) for x in rng.poisson(mask_length, size=num_mask)])
        else:
            raise Exception("unknown mask selection " + mask_type)
        if lengths.sum() == 0:
            lengths[0] = min(mask_length, sz - 1)
        min_len = int(lengths.min())
        if sz - min_len <= num_mask:
            min_len = sz - num_mask - 1
        starts = rng.choice(sz - min_len, num_mask, replace=False)
        # span expansion, vectorised: starts[j] + 0..lengths[j]-1
        reps = np.repeat(starts, lengths)
        offs = np.arange(int(lengths.sum())) - np.repeat(np.cumsum(lengths) - lengths, lengths)
        idc = reps + offs
        rows.append(np.unique(idc[idc < sz]))
    min_len = min(len(r) for r in rows)
    mask = np.zeros((bsz, all_sz), dtype=bool)
    for i, idc in enumerate(rows):
        if len(idc) > min_len:
            idc = rng.choice(idc, min_len, replace=False)
        mask[i, idc] = True
    return mask


def sample_negative_indices(bsz: int, num: int, n_negatives: int) -> torch.Tensor:
    """Index tensor of Wav2Vec2Model.sample_negatives (fs/models/wav2vec/wav2vec2.py:484-495,
    512-514): one torch.randint on the CPU default generator; never the positive itself; indices
    address the flattened (bsz*num) target rows.  int64 (bsz, n_negatives*num)."""
    assert num > 1, f"{bsz, num}"
    own = torch.arange(num).repeat_interleave(n_negatives)
    neg = torch.randint(low=0, high=num - 1, size=(bsz, n_negatives * num))
    neg += (neg >= own).to(neg.dtype)
    neg += (torch.arange(bsz) * num).unsqueeze(1)
    return neg


def sample_context(context_type: str, main_context: int, right_context: int) -> Tuple[int, int]:
    """fs/models/wav2vec/wav2vec_S.py:392-404."""
    if context_type == "sampling":
        m = random.randint(4, 16) * 2
        r = random.randint(2, 8) * 2
        return m, min(r, m // 2)
    if context_type == "constant":
        return main_context, right_context
    raise ValueError("The mode of context_type: ({}) cannot be used. Please check.".format(context_type))


def layerdrop_keep(num_layers: int, layerdrop: float, training: bool) -> List[bool]:
    """One np.random.random() per layer when layerdrop > 0 (wav2vec_S.py:414-416); note the
    draw happens in eval mode too, exactly as in the reference."""
    keep = []
    for _ in range(num_layers):
        p = np.random.random() if layerdrop > 0 else 1
        keep.append((not training) or (p > layerdrop))
    return keep


class BlockLayout:
    """Integer structure of gen_block_attn_mask (wav2vec_S.py:444-489) for one (T', m, r):
    which frame every token row copies, which copies each frame has (CSR, for the backward
    gather), and the key-padding flags of the appended rows."""

    def __init__(self, Tp: int, m: int, r: int):
        self.Tp, self.m, self.r = Tp, m, r
        nb = Tp // m
        if r > 0 and nb > 0:
            rc = ((np.arange(nb)[:, None] + 1) * m + np.arange(r)[None, :]).reshape(-1)
            self.rc_oob = rc > Tp - 1
            self.rc_idx = np.clip(rc, 0, Tp - 1)
        else:
            self.rc_idx = np.zeros(0, dtype=np.int64)
            self.rc_oob = np.zeros(0, dtype=bool)
        self.R = len(self.rc_idx)
        self.N = Tp + self.R
        self.src = np.concatenate([np.arange(Tp), self.rc_idx]).astype(np.int32)
        order = np.argsort(self.rc_idx, kind="stable")
        counts = np.bincount(self.rc_idx, minlength=Tp) if self.R else np.zeros(Tp, dtype=np.int64)
        self.copy_start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        self.copy_list = (Tp + order).astype(np.int32) if self.R else np.zeros(1, dtype=np.int32)

    def key_padding(self, pad: Optional[np.ndarray], B: int) -> Optional[np.ndarray]:
        """pad: [B, T'] bool (or None).  Returns uint8 [B, N] or None when nothing is padded."""
        if pad is None and not self.rc_oob.any():
            return None
        if pad is None:
            pad = np.zeros((B, self.Tp), dtype=bool)
        full = np.concatenate([pad, pad[:, self.rc_idx] | self.rc_oob[None, :]], axis=1)
        return full.astype(np.uint8)


_LAYOUTS = {}


def block_layout(Tp: int, m: int, r: int) -> BlockLayout:
    key = (Tp, m, r)
    if key not in _LAYOUTS:
        if len(_LAYOUTS) > 256:
            _LAYOUTS.clear()
        _LAYOUTS[key] = BlockLayout(Tp, m, r)
    return _LAYOUTS[key]


def sinusoidal_table(num_embeddings: int, dim: int, padding_idx: int = 1) -> torch.Tensor:
    """fs/modules/sinusoidal_positional_embedding.py:35-58 (sin half | cos half, padding row zero)."""
    import math
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
    ang = torch.arange(num_embeddings, dtype=torch.float).unsqueeze(1) * freq.unsqueeze(0)
    emb = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1).view(num_embeddings, -1)
    if dim % 2 == 1:
        emb = torch.cat([emb, torch.zeros(num_embeddings, 1)], dim=1)
    emb[padding_idx, :] = 0
    return emb


def positions_from_padding(pad: Optional[torch.Tensor], B: int, T: int, padding_idx: int = 1) -> torch.Tensor:
    """fs/utils.py:250-260 on the bool padding mask (wav2vec_S.py:357-367): int32 [B, T]."""
    if pad is None:
        return (torch.arange(T, dtype=torch.int32) + padding_idx + 1).unsqueeze(0).expand(B, T).contiguous()
    nonpad = (~pad.bool()).int().cpu()
    return ((torch.cumsum(nonpad, dim=1) * nonpad) + padding_idx).int().contiguous()
