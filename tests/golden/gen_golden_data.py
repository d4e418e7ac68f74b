#!/usr/bin/env python3
"""Golden vectors for SURVEY.md section 8 row f3 (input side), recorded from the REAL reference in the build
container: the compiled Cython batcher (oracle/_ref, built by `make -C oracle ref` from
fairseq/fairseq/data/data_utils_fast.pyx) and fs/data/audio/raw_audio_dataset.py (oracle/ref_import.load_data).

    make -C oracle ref && python tests/golden/gen_golden_data.py

Output (data only): tests/golden/data_side.npz
    bbs.<i>.*      batch_by_size_vec cases: sizes, (max_tokens, max_sentences, bsz_mult), batch ends
    order.*        ordered_indices for seeded shuffles
    collate.<i>.*  collater cases (crop-to-min, pad-to-max, max_sample_size crop, normalize): inputs, seeds, outputs
    manifest.*     FileAudioDataset manifest parsing (tsv text in, kept names / sizes / skipped out), bucket sizes
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_import  # noqa: E402

D = ref_import.load_data()
assert D.data_utils_fast is not None, "build the reference batcher first: make -C oracle ref"


def librispeech_like_sizes(rng, n):
    """Utterance lengths shaped like LibriSpeech (2-35 s at 16 kHz, mode ~14 s)."""
    sec = np.clip(rng.gamma(6.0, 2.3, size=n), 1.3, 35.0)
    return (sec * 16000).astype(np.int64)


def gen_bbs(out):
    rng = np.random.RandomState(7)
    cases = [
        # (n, max_tokens, max_sentences, bsz_mult, kind)
        (10, 10, -1, 1, "tiny"), (200, 1400000, -1, 1, "libri_sorted"), (200, 1400000, -1, 8, "libri_sorted"),
        (500, 1400000, 6, 1, "libri_sorted"), (300, 1400000, -1, 8, "libri_unsorted"), (64, -1, 5, 2, "libri_sorted"),
        (257, 400000, -1, 4, "libri_sorted_capped"), (1, 1400000, -1, 8, "libri_sorted"), (97, 1200000, 12, 8, "libri_unsorted"),
        (1200, 1400000, -1, 8, "libri_sorted_capped"),
    ]
    for i, (n, mt, ms, mult, kind) in enumerate(cases):
        if kind == "tiny":
            sizes = np.array([5, 4, 4, 3, 3, 3, 2, 2, 1, 1], dtype=np.int64)
        else:
            sizes = librispeech_like_sizes(rng, n)
            if "capped" in kind:
                sizes = np.minimum(sizes, 250000)               # size() = min(size, max_sample_size), :206-211
            if mt > 0:
                sizes = np.minimum(sizes, mt)
            if "unsorted" not in kind:
                sizes = np.sort(sizes)[::-1].copy()
        idx = rng.permutation(n).astype(np.int64)
        batches = D.data_utils_fast.batch_by_size_vec(idx, sizes, mt, ms, mult)
        ends = np.cumsum([len(b) for b in batches]).astype(np.int32)
        assert np.array_equal(np.concatenate(batches), idx)
        out[f"bbs.{i}.sizes"] = sizes
        out[f"bbs.{i}.indices"] = idx
        out[f"bbs.{i}.args"] = np.array([mt, ms, mult], dtype=np.int64)
        out[f"bbs.{i}.ends"] = ends
    out["bbs.count"] = np.array([len(cases)])


class MemDataset(D.RawAudioDataset):
    def __init__(self, waves, **kw):
        super().__init__(sample_rate=16000, **kw)
        self.waves = waves
        self.sizes = [len(w) for w in waves]

    def __getitem__(self, i):
        return {"id": i, "source": self.postprocess(self.waves[i].clone(), 16000)}


def gen_order_collate(out):
    rng = np.random.RandomState(11)
    sizes = librispeech_like_sizes(rng, 300)
    sizes[::7] = sizes[3]                                        # ties: the permutation key decides
    ds = MemDataset([torch.zeros(1)] * 0, shuffle=True)
    ds.sizes = sizes.tolist()
    for seed in (0, 1):
        np.random.seed(seed)
        out[f"order.s{seed}"] = np.ascontiguousarray(ds.ordered_indices()).astype(np.int64)
        out[f"order.s{seed}.next"] = np.array([np.random.rand()])
    ds.shuffle = False
    out["order.noshuffle"] = np.ascontiguousarray(ds.ordered_indices()).astype(np.int64)
    out["order.sizes"] = sizes
    g = torch.Generator().manual_seed(5)
    cases = [
        dict(lens=[4000, 3111, 5200, 3111], pad=False, max_sample_size=250000, normalize=False),
        dict(lens=[4000, 3111, 5200], pad=False, max_sample_size=3000, normalize=False),
        dict(lens=[4000, 3111, 5200, 777], pad=True, max_sample_size=250000, normalize=False),
        dict(lens=[4000, 3111, 5200, 777], pad=True, max_sample_size=3500, normalize=True),
        dict(lens=[9000, 20000, 8192, 8193], pad=False, max_sample_size=250000, normalize=True),
    ]
    for i, c in enumerate(cases):
        waves = [torch.randn(n, generator=g) * (0.05 + 0.02 * j) + 0.01 * j for j, n in enumerate(c["lens"])]
        ds = MemDataset(waves, max_sample_size=c["max_sample_size"], pad=c["pad"], normalize=c["normalize"])
        np.random.seed(100 + i)
        res = ds.collater([ds[j] for j in range(len(waves))])
        out[f"collate.{i}.next"] = np.array([np.random.rand()])
        out[f"collate.{i}.flat"] = torch.cat(waves).numpy()
        out[f"collate.{i}.lens"] = np.array(c["lens"])
        out[f"collate.{i}.args"] = np.array([int(c["pad"]), c["max_sample_size"], int(c["normalize"])])
        out[f"collate.{i}.source"] = res["net_input"]["source"].numpy()
        out[f"collate.{i}.id"] = res["id"].numpy()
        if c["pad"]:
            out[f"collate.{i}.padding_mask"] = res["net_input"]["padding_mask"].numpy()
    out["collate.count"] = np.array([len(cases)])


def gen_manifest(out):
    rng = np.random.RandomState(3)
    sizes = librispeech_like_sizes(rng, 40)
    sizes[5] = 20000
    sizes[17] = 31999
    lines = ["/data/LibriSpeech/train-clean-100"] + [f"{100 + i}/{i}/{100 + i}-{i}-0001.flac\t{int(s)}" for i, s in enumerate(sizes)]
    text = "\n".join(lines) + "\n"
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "train.tsv")
    open(path, "w").write(text)
    ds = D.FileAudioDataset(path, sample_rate=16000, max_sample_size=250000, min_sample_size=32000, pad=True, num_buckets=4)
    out["manifest.text"] = np.frombuffer(text.encode(), dtype=np.uint8)
    out["manifest.sizes"] = np.array(ds.sizes)
    out["manifest.fnames"] = np.frombuffer("\n".join(ds.fnames).encode(), dtype=np.uint8)
    out["manifest.line_inds"] = np.array(sorted(ds.line_inds))
    out["manifest.root"] = np.frombuffer(ds.root_dir.encode(), dtype=np.uint8)
    out["manifest.buckets"] = np.asarray(ds.buckets)
    out["manifest.bucketed_sizes"] = np.asarray(ds._bucketed_sizes)
    out["manifest.size_of"] = np.array([ds.size(i) for i in range(len(ds))])


if __name__ == "__main__":
    out = {}
    gen_bbs(out)
    gen_order_collate(out)
    gen_manifest(out)
    np.savez_compressed(os.path.join(HERE, "data_side.npz"), **out)
    print("data_side.npz", len(out), "arrays")
