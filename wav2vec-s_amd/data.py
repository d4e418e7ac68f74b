"""Input side of the hot path: what stands between a LibriSpeech manifest and
``model(source=...)``.

Host-side mirror of
* ``fs/data/audio/raw_audio_dataset.py``: ``RawAudioDataset`` (:20-230) / ``FileAudioDataset`` (:233-330) - manifest
  ``.tsv`` format, ``min_sample_size`` filter, ``size / num_tokens``, ``ordered_indices`` (shuffle + length sort on
  the numpy global generator), ``collater`` (crop-to-min or pad-to-max, crop offsets from ``np.random.randint``),
  per-utterance normalisation, bucket sizes;
* ``fs/data/data_utils.py::batch_by_size`` (:281-355) over the Cython ``batch_by_size_vec``
  (``fs/data/data_utils_fast.pyx:19-98``) - here the C ABI host function ``w2vs_batch_by_size``;
* ``FairseqDataset.batch_by_size`` (``fs/data/fairseq_dataset.py:104-153``).

What differs by design: the collater does its arithmetic on the GPU.  ``__getitem__`` returns the utterance as read
from disk; the batch goes to the device as ONE flat pinned buffer and ``w2vs_collate`` normalises (statistics over the
whole utterance, as ``postprocess`` does before any crop), crops, pads, builds the padding mask and casts to the
model's dtype.  Host RNG draws (shuffle permutation, crop offsets) are made with the same numpy calls in the same
order as the reference, so a seeded run selects the same samples and the same crops.
"""
import ctypes as C
import os
import sys
import wave
from typing import List

import numpy as np
import torch

from . import _lib
from ._lib import W2vsError

BF16 = torch.bfloat16


# ---------------------------------------------------------------------------------------------- batching (host, C ABI)
def batch_by_size_vec(indices: np.ndarray, num_tokens_vec: np.ndarray, max_tokens: int, max_sentences: int,
                      bsz_mult: int) -> List[np.ndarray]:
    """data_utils_fast.pyx:19-98 through ``w2vs_batch_by_size``; returns the list of index arrays."""
    indices = np.ascontiguousarray(indices, dtype=np.int64)
    sizes = np.ascontiguousarray(num_tokens_vec, dtype=np.int64)
    n = int(indices.shape[0])
    if n == 0:
        return []
    if sizes.shape[0] != n:
        raise W2vsError("batch_by_size: indices and num_tokens_vec differ in length")
    ends = np.zeros(n + 1, dtype=np.int32)
    nb = C.c_int32(0)
    _lib.call("w2vs_batch_by_size", sizes.ctypes.data, n, int(max_tokens), int(max_sentences), int(bsz_mult),
              ends.ctypes.data, C.addressof(nb))
    return np.split(indices, ends[:nb.value - 1])


def batch_by_size(indices, num_tokens_fn, num_tokens_vec=None, max_tokens=None, max_sentences=None,
                  required_batch_size_multiple=1, fixed_shapes=None):
    """fs/data/data_utils.py:281-355 (same argument meaning; ``fixed_shapes`` is a TPU feature and is not built)."""
    if fixed_shapes is not None:
        raise W2vsError("batch_by_size(fixed_shapes=...) (TPU batch shapes) is not built")
    max_tokens = int(max_tokens) if max_tokens is not None else -1
    max_sentences = max_sentences if max_sentences is not None else -1
    if not isinstance(indices, np.ndarray):
        indices = np.fromiter(indices, dtype=np.int64, count=-1)
    if num_tokens_vec is None:                                  # batch_by_size_fn, data_utils_fast.pyx:101-118
        num_tokens_vec = np.fromiter((num_tokens_fn(int(i)) for i in indices), dtype=np.int64, count=len(indices))
    elif not isinstance(num_tokens_vec, np.ndarray):
        num_tokens_vec = np.fromiter(num_tokens_vec, dtype=np.int64, count=-1)
    if max_tokens > 0 and len(num_tokens_vec) and int(np.max(num_tokens_vec)) > max_tokens:
        raise AssertionError(f"Sentences lengths should not exceed max_tokens={max_tokens}")   # :30-32
    return batch_by_size_vec(indices, num_tokens_vec, max_tokens, max_sentences, required_batch_size_multiple)


def get_buckets(sizes, num_buckets):
    """fs/data/data_utils.py:541-549."""
    return np.unique(np.percentile(sizes, np.linspace(0, 100, num_buckets + 1), method="lower")[1:])


def get_bucketed_sizes(orig_sizes, buckets):
    """fs/data/data_utils.py:552-560."""
    sizes = np.copy(orig_sizes)
    assert np.min(sizes) >= 0
    start_val = -1
    for end_val in buckets:
        sizes[(sizes > start_val) & (sizes <= end_val)] = end_val
        start_val = end_val
    return sizes


# ---------------------------------------------------------------------------------------------- collate (device)
class _Staging:
    """A ring of pinned host buffers for the batch's samples and index arrays.  A slot is reused only after the H2D
    copies that read it have completed (an event per slot), so back-to-back collater calls neither race on the buffer
    nor make the pinned allocator grow (hipHostMalloc costs 100+ ms)."""

    def __init__(self, slots=3):
        self.slots = [None] * slots
        self.next = 0

    def take(self, n_float, n_int, dev):
        i = self.next
        self.next = (i + 1) % len(self.slots)
        slot = self.slots[i]
        if slot is not None and slot[2] is not None:
            slot[2].synchronize()
        if slot is None or slot[0].numel() < n_float or slot[1].numel() < n_int:
            cap_f = max(n_float, 0 if slot is None else slot[0].numel())
            cap_i = max(n_int, 64, 0 if slot is None else slot[1].numel())
            slot = [torch.empty(int(cap_f * 1.25) + 1, dtype=torch.float32, pin_memory=True),
                    torch.empty(cap_i, dtype=torch.int32, pin_memory=True), None]
            self.slots[i] = slot
        return slot[0], slot[1], i

    def done(self, i, dev):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self.slots[i][2] = ev


_STAGING = _Staging()


def collate_device(sources: List[torch.Tensor], target_size: int, crop_starts, *, pad: bool, normalize: bool,
                   device, dtype=BF16, num_pad: int = 0):
    """One batch through ``w2vs_collate``.  sources: 1-D fp32 CPU tensors as read from disk; crop_starts[b]: first kept
    sample of an utterance longer than ``target_size``.  Returns (source [B, target(+num_pad)], padding_mask or None);
    ``num_pad`` extra columns (bucketing, :157-167) are zeros with padding_mask True."""
    if dtype not in (BF16, torch.float32):
        raise W2vsError("collate: output dtype must be bfloat16 or float32")
    dev = torch.device(device)
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise W2vsError("collate runs on an MI355X only (there is no CPU path)")
    B = len(sources)
    sizes = np.array([int(s.numel()) for s in sources], dtype=np.int32)
    offs = np.zeros(B, dtype=np.int64)
    offs[1:] = np.cumsum(sizes[:-1].astype(np.int64))
    total = int(sizes.sum())
    flat, meta, slot = _STAGING.take(max(total, 1), 4 * B, dev)
    for s, o, n in zip(sources, offs, sizes):
        flat[o:o + n] = s.reshape(-1)
    mnp = meta.numpy()                                             # offsets (as 2 x int32), sizes, crop starts
    mnp[:2 * B] = offs.view(np.int32)
    mnp[2 * B:3 * B] = sizes
    mnp[3 * B:4 * B] = np.asarray(crop_starts, dtype=np.int32)
    flat_d = flat[:max(total, 1)].to(dev, non_blocking=True)
    meta_d = meta[:4 * B].to(dev, non_blocking=True)
    _STAGING.done(slot, dev)
    width = target_size + num_pad
    out = torch.empty(B, width, dtype=dtype, device=dev)
    pm = torch.empty(B, width, dtype=torch.uint8, device=dev) if pad else None
    max_size = int(sizes.max()) if B else 0
    lib = _lib.load()
    nch = int(lib.w2vs_collate_chunks(max_size))
    partial = torch.empty(B * nch * 2, dtype=torch.float64, device=dev) if normalize else None
    d = _lib.CollateDesc()
    d.flat = flat_d.data_ptr()
    d.offset = meta_d.data_ptr()
    d.size = meta_d.data_ptr() + 8 * B
    d.crop_start = meta_d.data_ptr() + 12 * B
    d.out = out.data_ptr()
    d.padding_mask = pm.data_ptr() if pm is not None else None
    d.partial = partial.data_ptr() if partial is not None else None
    d.B, d.target, d.width, d.max_size = B, target_size, width, max_size
    d.normalize, d.out_f32 = int(bool(normalize)), int(dtype == torch.float32)
    _lib.call("w2vs_collate", C.byref(d), torch.cuda.current_stream(dev).cuda_stream)
    return out, (pm.bool() if pm is not None else None)


# ---------------------------------------------------------------------------------------------- datasets
class RawAudioDataset(torch.utils.data.Dataset):
    """fs/data/audio/raw_audio_dataset.py:20-230."""

    def __init__(self, sample_rate, max_sample_size=None, min_sample_size=0, shuffle=True, pad=False, normalize=False,
                 compute_mask_indices=False, device="cuda", dtype=BF16, **mask_compute_kwargs):
        super().__init__()
        if compute_mask_indices:
            raise W2vsError("compute_mask_indices=True (TPU: masks precomputed in the collater) is not built; the "
                            "model draws its masks itself (wav2vec2.py:452-469)")
        self.sample_rate = sample_rate
        self.sizes = []
        self.max_sample_size = max_sample_size if max_sample_size is not None else sys.maxsize
        self.min_sample_size = min_sample_size
        self.pad = pad
        self.shuffle = shuffle
        self.normalize = normalize
        self.device, self.dtype = device, dtype
        self.num_buckets = 0

    def __getitem__(self, index):
        raise NotImplementedError()

    def __len__(self):
        return len(self.sizes)

    def postprocess(self, feats, curr_sample_rate):
        """:60-72 without the normalisation, which ``collater`` applies on the GPU (over the whole utterance)."""
        if feats.dim() == 2:
            feats = feats.mean(-1)
        if curr_sample_rate != self.sample_rate:
            raise Exception(f"sample rate: {curr_sample_rate}, need {self.sample_rate}")
        assert feats.dim() == 1, feats.dim()
        return feats

    def crop_start(self, size, target_size):
        """The draw of crop_to_max_size (:73-81): np.random.randint(0, diff + 1) only when the utterance is longer."""
        diff = size - target_size
        if diff <= 0:
            return 0
        return int(np.random.randint(0, diff + 1))

    def collater(self, samples):
        """:123-192.  Same keys and shapes; ``source`` (and ``padding_mask``) are device tensors."""
        samples = [s for s in samples if s["source"] is not None]
        if len(samples) == 0:
            return {}
        sources = [s["source"] for s in samples]
        sizes = [len(s) for s in sources]
        if self.pad:
            target_size = min(max(sizes), self.max_sample_size)
        else:
            target_size = min(min(sizes), self.max_sample_size)
        starts = [self.crop_start(size, target_size) for size in sizes]       # reference order: one draw per long row
        num_pad = 0
        if self.num_buckets > 0:
            assert self.pad, "Cannot bucket without padding first."
            bucket = max(self._bucketed_sizes[s["id"]] for s in samples)
            num_pad = int(bucket - target_size)
        src, pm = collate_device(sources, target_size, starts, pad=self.pad, normalize=self.normalize,
                                 device=self.device, dtype=self.dtype, num_pad=num_pad)
        inp = {"source": src}
        out = {"id": torch.LongTensor([s["id"] for s in samples])}
        if self.pad:
            inp["padding_mask"] = pm
        out["net_input"] = inp
        return out

    def num_tokens(self, index):
        return self.size(index)

    def size(self, index):
        if self.pad:
            return self.sizes[index]
        return min(self.sizes[index], self.max_sample_size)

    def ordered_indices(self):
        """:214-224."""
        if self.shuffle:
            order = [np.random.permutation(len(self))]
        else:
            order = [np.arange(len(self))]
        order.append(self.sizes)
        return np.lexsort(order)[::-1]

    def batch_by_size(self, indices, max_tokens=None, max_sentences=None, required_batch_size_multiple=1):
        """fs/data/fairseq_dataset.py:104-153 (no fixed shapes)."""
        return batch_by_size(indices, num_tokens_fn=self.num_tokens, num_tokens_vec=None, max_tokens=max_tokens,
                             max_sentences=max_sentences, required_batch_size_multiple=required_batch_size_multiple)

    def filter_indices_by_size(self, indices, max_sizes):
        """fs/data/fairseq_dataset.py:155-190 for a scalar limit."""
        sz = np.array([self.size(int(i)) for i in indices])
        keep = sz <= max_sizes
        return indices[keep], indices[~keep].tolist()

    def set_bucket_info(self, num_buckets):
        """:268-287."""
        self.num_buckets = num_buckets
        if self.num_buckets > 0:
            self._collated_sizes = np.minimum(np.array(self.sizes), self.max_sample_size)
            self.buckets = get_buckets(self._collated_sizes, self.num_buckets)
            self._bucketed_sizes = get_bucketed_sizes(self._collated_sizes, self.buckets)


def read_manifest(manifest_path, min_sample_size=None):
    """The ``.tsv`` format of examples/wav2vec/wav2vec_manifest.py as FileAudioDataset.__init__ parses it (:256-267):
    first line = root directory, then ``relative/path<TAB>num_samples``.  Returns (root, fnames, sizes, line_inds,
    skipped)."""
    fnames, sizes, line_inds, skipped = [], [], set(), 0
    with open(manifest_path, "r") as f:
        root_dir = f.readline().strip()
        for i, line in enumerate(f):
            items = line.strip().split("\t")
            assert len(items) == 2, line
            sz = int(items[1])
            if min_sample_size is not None and sz < min_sample_size:
                skipped += 1
                continue
            fnames.append(items[0])
            line_inds.add(i)
            sizes.append(sz)
    return root_dir, fnames, sizes, line_inds, skipped


def read_audio(fname):
    """(samples float32 in [-1, 1) [n] or [n, channels], sample_rate).  ``soundfile`` when it is installed (what the
    reference uses, :292-295); otherwise PCM ``.wav`` through the standard library."""
    try:
        import soundfile as sf
        wav, sr = sf.read(fname)
        return torch.from_numpy(wav).float(), sr
    except ImportError:
        pass
    if not fname.lower().endswith(".wav"):
        raise W2vsError(f"reading {fname} needs the soundfile package (only PCM .wav is read without it)")
    with wave.open(fname, "rb") as w:
        sr, nch, width, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if width == 2:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        a = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    else:
        raise W2vsError(f"{fname}: unsupported PCM sample width {width}")
    if nch > 1:
        a = a.reshape(-1, nch)
    return torch.from_numpy(a.copy()), sr


class FileAudioDataset(RawAudioDataset):
    """fs/data/audio/raw_audio_dataset.py:233-330."""

    def __init__(self, manifest_path, sample_rate, max_sample_size=None, min_sample_size=0, shuffle=True, pad=False,
                 normalize=False, num_buckets=0, compute_mask_indices=False, device="cuda", dtype=BF16,
                 **mask_compute_kwargs):
        super().__init__(sample_rate=sample_rate, max_sample_size=max_sample_size, min_sample_size=min_sample_size,
                         shuffle=shuffle, pad=pad, normalize=normalize, compute_mask_indices=compute_mask_indices,
                         device=device, dtype=dtype, **mask_compute_kwargs)
        self.root_dir, self.fnames, self.sizes, self.line_inds, self.skipped = read_manifest(manifest_path, min_sample_size)
        self.set_bucket_info(num_buckets)

    def __getitem__(self, index):
        fname = os.path.join(self.root_dir, self.fnames[index])
        wav, sr = read_audio(fname)
        return {"id": index, "source": self.postprocess(wav, sr)}
