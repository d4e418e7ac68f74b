"""Row f3 on the GPU: the device collater (``w2vs_collate``: whole-utterance normalisation, crop, pad, padding mask,
cast) against the vectors recorded from the reference's RawAudioDataset.collater and against the oracle, plus
size-independent properties at the pre-training batch shape.  Needs an MI355X: pytest -m gpu"""
import os
import wave

import numpy as np
import pytest
import torch

import w2vs_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def z(golden_dir):
    return np.load(os.path.join(golden_dir, "data_side.npz"))


class Mem:
    """An in-memory dataset on the product's RawAudioDataset."""

    def __new__(cls, waves, **kw):
        from wav2vec_s_amd import data

        class _D(data.RawAudioDataset):
            def __getitem__(self, i):
                return {"id": i, "source": self.postprocess(self.waves[i].clone(), 16000)}
        d = _D(16000, **kw)
        d.waves = waves
        d.sizes = [len(w) for w in waves]
        return d


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_collater_matches_reference_vectors(z, dtype):
    for i in range(int(z["collate.count"][0])):
        pad, max_size, norm = (int(v) for v in z[f"collate.{i}.args"])
        waves = list(torch.tensor(z[f"collate.{i}.flat"]).split(z[f"collate.{i}.lens"].tolist()))
        ds = Mem(waves, max_sample_size=max_size, pad=bool(pad), normalize=bool(norm), dtype=dtype)
        np.random.seed(100 + i)
        res = ds.collater([ds[j] for j in range(len(waves))])
        assert np.random.rand() == z[f"collate.{i}.next"][0]                      # same crop draws, same order
        src = res["net_input"]["source"]
        want = torch.tensor(z[f"collate.{i}.source"])
        assert src.is_cuda and src.dtype == dtype and tuple(src.shape) == tuple(want.shape)
        assert np.array_equal(res["id"].numpy(), z[f"collate.{i}.id"])
        if dtype == torch.float32:
            np.testing.assert_allclose(src.cpu().numpy(), want.numpy(), atol=2e-6, rtol=2e-6)
        else:                                                                        # RNE cast of the same fp32 value
            assert float((src.float().cpu() - want.to(dtype).float()).abs().max()) <= float(want.abs().max()) * 2 ** -8
            assert float((src.float().cpu() != want.to(dtype).float()).float().mean()) < 0.02
        if pad:
            pm = res["net_input"]["padding_mask"]
            assert pm.dtype == torch.bool and np.array_equal(pm.cpu().numpy(), z[f"collate.{i}.padding_mask"])
        else:
            assert "padding_mask" not in res["net_input"]


def test_collater_pretraining_batch_shape_properties():
    """8 utterances of 11-16 s cropped to the shortest (the pre-training collater: pad=False), normalised:
    every row is a contiguous slice of its normalised utterance; per-utterance statistics use ALL samples."""
    g = torch.Generator().manual_seed(3)
    lens = [250000, 176000, 243111, 175000, 201234, 199999, 180001, 250000]
    waves = [torch.randn(n, generator=g) * (0.03 * (j + 1)) + 0.1 * j for j, n in enumerate(lens)]
    ds = Mem(waves, max_sample_size=250000, pad=False, normalize=True, dtype=torch.float32)
    np.random.seed(9)
    res = ds.collater([ds[j] for j in range(8)])
    src = res["net_input"]["source"].cpu()
    assert tuple(src.shape) == (8, 175000)
    np.random.seed(9)
    want, _, starts = O.collate(waves, pad=False, max_sample_size=250000, normalize=True)
    np.testing.assert_allclose(src.numpy(), want.numpy(), atol=3e-5, rtol=1e-5)
    for j in range(8):
        full = (waves[j].double() - waves[j].double().mean()) / torch.sqrt(waves[j].double().var(unbiased=False) + 1e-5)
        np.testing.assert_allclose(src[j].numpy(), full[starts[j]:starts[j] + 175000].float().numpy(), atol=3e-5)
    assert starts[3] == 0 and max(starts) > 0


def test_bucketed_padding_and_file_dataset_end_to_end(tmp_path):
    """Manifest -> PCM .wav files -> ordered_indices -> batch_by_size -> device collater (padded + bucketed) ->
    the streaming twin's padded forward.  Exercises the whole input side in front of the model."""
    from wav2vec_s_amd import data, streaming
    import argparse
    g = torch.Generator().manual_seed(1)
    lens = [12000, 9000, 16000, 7000, 15000, 3000, 11000]
    os.makedirs(os.path.join(tmp_path, "a"))
    lines = [str(tmp_path)]
    for i, n in enumerate(lens):
        pcm = (torch.randn(n, generator=g).clamp(-3, 3) * 3000).to(torch.int16).numpy()
        with wave.open(os.path.join(tmp_path, "a", f"{i}.wav"), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
        lines.append(f"a/{i}.wav\t{n}")
    open(os.path.join(tmp_path, "train.tsv"), "w").write("\n".join(lines) + "\n")
    ds = data.FileAudioDataset(os.path.join(tmp_path, "train.tsv"), 16000, max_sample_size=250000, min_sample_size=5000,
                               pad=True, normalize=True, num_buckets=2, dtype=torch.bfloat16)
    assert len(ds) == 6 and ds.skipped == 1
    np.random.seed(0)
    order = ds.ordered_indices()
    batches = ds.batch_by_size(order, max_tokens=40000, required_batch_size_multiple=1)
    assert sorted(np.concatenate(batches).tolist()) == list(range(6))
    for b in batches:
        assert len(b) * max(ds.size(int(i)) for i in b) <= 40000
    b = batches[0]
    res = ds.collater([ds[int(i)] for i in b])
    src, pm = res["net_input"]["source"], res["net_input"]["padding_mask"]
    bucket = max(ds._bucketed_sizes[int(i)] for i in b)
    assert tuple(src.shape) == (len(b), bucket) and src.dtype == torch.bfloat16 and pm.shape == src.shape
    for row, i in enumerate(b):
        n = ds.sizes[int(i)]
        assert not pm[row, :n].any() and pm[row, n:].all()
        assert float(src[row, n:].abs().max() if n < bucket else 0.0) == 0.0
        x = src[row, :n].float()
        assert abs(float(x.mean())) < 2e-2 and abs(float(x.var(unbiased=False)) - 1.0) < 3e-2
    kw = dict(extractor_mode="layer_norm", encoder_layers=2, encoder_embed_dim=128, encoder_ffn_embed_dim=256,
              encoder_attention_heads=2, conv_feature_layers="[(64, 10, 5)] + [(64, 3, 2)] * 4 + [(64,2,2)] * 2",
              main_context=8, right_context=4, pos_type="sin", load_pretrained_model_from=None)
    model = streaming.BlockWiseWav2Vec2Model.build_model(argparse.Namespace(**kw)).to(torch.bfloat16).cuda().eval()
    with torch.no_grad():
        out = model(src, pm)
    x, fp = out["encoder_out"][0], out["encoder_padding_mask"][0]
    assert x.shape[1] == len(b) and fp.shape == (len(b), x.shape[0]) and bool(torch.isfinite(x.float()).all())
