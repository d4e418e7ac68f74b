"""Row f3 (input side) without a GPU: the oracle restatements against the vectors recorded from the real reference
(tests/golden/data_side.npz; live against the compiled reference Cython when oracle/_ref holds it), and the product's
host pieces - the C ABI batcher ``w2vs_batch_by_size``, ordering, manifest parsing, buckets - bit-exact against both."""
import os

import numpy as np
import pytest
import torch

import ref_import
import w2vs_oracle as O


@pytest.fixture(scope="module")
def z(golden_dir):
    return np.load(os.path.join(golden_dir, "data_side.npz"))


def _ends(batches):
    return np.cumsum([len(b) for b in batches]).astype(np.int32)


def test_oracle_batcher_matches_reference_vectors(z):
    for i in range(int(z["bbs.count"][0])):
        mt, ms, mult = (int(v) for v in z[f"bbs.{i}.args"])
        got = O.batch_by_size_vec(z[f"bbs.{i}.indices"], z[f"bbs.{i}.sizes"], mt, ms, mult)
        assert np.array_equal(_ends(got), z[f"bbs.{i}.ends"]), i
        assert np.array_equal(np.concatenate(got), z[f"bbs.{i}.indices"])


def test_product_batcher_matches_reference_vectors(z):
    from wav2vec_s_amd import data
    for i in range(int(z["bbs.count"][0])):
        mt, ms, mult = (int(v) for v in z[f"bbs.{i}.args"])
        got = data.batch_by_size_vec(z[f"bbs.{i}.indices"], z[f"bbs.{i}.sizes"], mt, ms, mult)
        assert np.array_equal(_ends(got), z[f"bbs.{i}.ends"]), i
        assert all(b.dtype == np.int64 for b in got)
        assert np.array_equal(np.concatenate(got), z[f"bbs.{i}.indices"])
        sizes = z[f"bbs.{i}.sizes"]
        lo = 0
        for e in z[f"bbs.{i}.ends"]:                           # the batching contract itself
            n, longest = e - lo, int(sizes[lo:e].max())
            assert mt <= 0 or n * longest <= mt
            assert ms <= 0 or n <= ms
            lo = e


def test_product_batcher_random_cases_equal_oracle_and_reference():
    from wav2vec_s_amd import data
    fast = ref_import.load_data().data_utils_fast if ref_import.available() else None
    rng = np.random.RandomState(0)
    for case in range(300):
        n = int(rng.randint(1, 120))
        sizes = rng.randint(1, 60, size=n).astype(np.int64)
        if case % 3 == 0:
            sizes = np.sort(sizes)[::-1].copy()
        mt = int(rng.choice([-1, 60, 100, 250, 1000]))
        ms = int(rng.choice([-1, 1, 3, 8, 17]))
        mult = int(rng.choice([1, 2, 4, 8]))
        idx = rng.permutation(n).astype(np.int64)
        want = O.batch_by_size_vec(idx, sizes, mt, ms, mult)
        got = data.batch_by_size_vec(idx, sizes, mt, ms, mult)
        assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want)), (case, mt, ms, mult)
        if fast is not None:
            ref = fast.batch_by_size_vec(idx, sizes, mt, ms, mult)
            assert len(ref) == len(got) and all(np.array_equal(a, b) for a, b in zip(got, ref)), (case, mt, ms, mult)


def test_batch_by_size_front_end_and_errors():
    from wav2vec_s_amd import data
    from wav2vec_s_amd._lib import W2vsError
    sizes = [5, 4, 4, 3, 3, 3, 2, 2, 1, 1]
    got = data.batch_by_size(range(10), lambda i: sizes[i], max_tokens=10)
    assert [b.tolist() for b in got] == [[0, 1], [2, 3], [4, 5, 6], [7, 8, 9]]
    assert data.batch_by_size(np.zeros(0, dtype=np.int64), None, np.zeros(0, dtype=np.int64), max_tokens=10) == []
    got = data.batch_by_size(np.arange(10), None, sizes, max_sentences=4, required_batch_size_multiple=1)
    assert [len(b) for b in got] == [4, 4, 2]
    with pytest.raises(AssertionError, match="max_tokens"):
        data.batch_by_size(np.arange(3), None, np.array([5, 50, 5]), max_tokens=10)
    with pytest.raises(W2vsError):
        data.batch_by_size_vec(np.arange(3), np.array([5, 50, 5]), 10, -1, 1)      # the C ABI rejects it itself
    with pytest.raises(W2vsError):
        data.batch_by_size(np.arange(3), None, np.array([5, 5, 5]), max_tokens=10, fixed_shapes=[(1, 2)])


def test_ordered_indices_oracle_and_product(z):
    from wav2vec_s_amd import data
    sizes = z["order.sizes"]
    ds = data.RawAudioDataset(16000, device="cpu")
    ds.sizes = sizes.tolist()
    for seed in (0, 1):
        np.random.seed(seed)
        assert np.array_equal(O.ordered_indices(sizes, True), z[f"order.s{seed}"])
        assert np.random.rand() == z[f"order.s{seed}.next"][0]
        np.random.seed(seed)
        assert np.array_equal(ds.ordered_indices(), z[f"order.s{seed}"])
        assert np.random.rand() == z[f"order.s{seed}.next"][0]
    ds.shuffle = False
    assert np.array_equal(ds.ordered_indices(), z["order.noshuffle"])
    assert np.array_equal(O.ordered_indices(sizes, False), z["order.noshuffle"])
    got = sizes[z["order.s0"]]
    assert np.all(got[:-1] >= got[1:])                          # longest first


def test_collate_oracle_matches_reference_vectors(z):
    for i in range(int(z["collate.count"][0])):
        pad, max_size, norm = (int(v) for v in z[f"collate.{i}.args"])
        waves = list(torch.tensor(z[f"collate.{i}.flat"]).split(z[f"collate.{i}.lens"].tolist()))
        np.random.seed(100 + i)
        src, pm, _ = O.collate(waves, pad=bool(pad), max_sample_size=max_size, normalize=bool(norm))
        assert np.random.rand() == z[f"collate.{i}.next"][0]     # same number of crop draws
        np.testing.assert_allclose(src.numpy(), z[f"collate.{i}.source"], atol=1e-6)
        if pad:
            assert np.array_equal(pm.numpy(), z[f"collate.{i}.padding_mask"])
        else:
            assert pm is None


def test_manifest_and_buckets(z, tmp_path):
    from wav2vec_s_amd import data
    path = os.path.join(tmp_path, "train.tsv")
    open(path, "w").write(bytes(z["manifest.text"]).decode())
    ds = data.FileAudioDataset(path, sample_rate=16000, max_sample_size=250000, min_sample_size=32000, pad=True,
                               num_buckets=4, device="cpu")
    assert ds.root_dir == bytes(z["manifest.root"]).decode()
    assert ds.fnames == bytes(z["manifest.fnames"]).decode().split("\n")
    assert np.array_equal(np.array(ds.sizes), z["manifest.sizes"]) and ds.skipped == 2
    assert sorted(ds.line_inds) == z["manifest.line_inds"].tolist()
    assert np.array_equal(ds.buckets, z["manifest.buckets"])
    assert np.array_equal(ds._bucketed_sizes, z["manifest.bucketed_sizes"])
    assert [ds.size(i) for i in range(len(ds))] == z["manifest.size_of"].tolist()
    ds.pad = False
    assert ds.size(0) == min(ds.sizes[0], 250000) and ds.num_tokens(0) == ds.size(0)


def test_collater_has_no_cpu_path():
    from wav2vec_s_amd import data
    from wav2vec_s_amd._lib import W2vsError
    ds = data.RawAudioDataset(16000, device="cpu")
    with pytest.raises(W2vsError, match="MI355X"):
        ds.collater([{"id": 0, "source": torch.zeros(100)}, {"id": 1, "source": torch.zeros(80)}])
    with pytest.raises(W2vsError):
        data.RawAudioDataset(16000, compute_mask_indices=True)
    assert ds.collater([]) == {}
