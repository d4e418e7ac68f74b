#!/usr/bin/env python3
"""Generate the committed golden vectors by running the REAL reference
(/root/reference, imported through oracle/ref_import.py) in the build container.

    python tests/golden/gen_golden.py

Outputs (data only - inputs and expected outputs, no reference source):
    tests/golden/host_rng.npz     G1 mask indices, G2 negative indices, G3 block
                                  structures, G4 sinusoid rows
    tests/golden/tiny_base.npz    G5 tiny post-LN model (12 layers, layer_norm_num=1):
                                  state_dict, input, recorded host draws, selected
                                  intermediates, loss, all parameter gradients
    tests/golden/tiny_large.npz   same for the large-style branches (pre-LN,
                                  conv_bias, layer_norm_num=7, feature_grad_mult=1)
    tests/golden/tiny_layerdrop.npz  train-mode run with LayerDrop + sampled contexts
The reference cannot travel to the GPU box; these files can.
"""
import hashlib
import os
import random
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_import  # noqa: E402

ref = ref_import.load()
w2 = ref.wav2vec2


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def gen_host():
    out = {}
    # G1: compute_mask_indices (fs/data/data_utils.py:389-513) as apply_mask calls it
    for (B, T) in [(2, 499), (8, 546), (5, 781), (3, 999), (2, 49)]:
        for s in (0, 1, 2):
            np.random.seed(s)
            m = ref.compute_mask_indices((B, T), None, 0.65, 10, "static", 0, min_masks=2,
                                         no_overlap=False, min_space=1)
            out[f"mask_{B}_{T}_s{s}"] = np.packbits(m, axis=1)
            # state of the global RNG after the call pins the draw COUNT as well
            out[f"mask_{B}_{T}_s{s}_next"] = np.array([np.random.rand()])
    # with a padding mask (one extra rand() per row, data_utils.py:433-440)
    np.random.seed(3)
    pm = torch.zeros(3, 200, dtype=torch.bool)
    pm[1, 150:] = True
    pm[2, 90:] = True
    m = ref.compute_mask_indices((3, 200), pm, 0.65, 10, "static", 0, min_masks=2)
    out["mask_pad_3_200_s3"] = np.packbits(m, axis=1)
    # G2: sample_negatives index tensor (fs/models/wav2vec/wav2vec2.py:471-527)
    class _M:  # minimal carrier of the attributes sample_negatives reads
        n_negatives = 100
        cross_sample_negatives = 0
    for (B, M, s) in [(2, 20, 0), (2, 247, 1), (8, 245, 2)]:
        torch.manual_seed(s)
        y = torch.zeros(B, M, 1)
        _, idx = w2.Wav2Vec2Model.sample_negatives(_M(), y, M)
        if B * M <= 64:
            out[f"neg_{B}_{M}_s{s}"] = idx.numpy()
        out[f"neg_{B}_{M}_s{s}_sha"] = np.frombuffer(bytes.fromhex(sha(idx.numpy())), dtype=np.uint8)
        out[f"neg_{B}_{M}_s{s}_head"] = idx.numpy()[:, :32].copy()
    # G3: gen_block_attn_mask (fs/models/wav2vec/wav2vec_S.py:444-489)
    for (Tp, m, r) in [(500, 16, 8), (546, 16, 8), (34, 8, 4), (40, 32, 16), (10, 16, 8), (50, 8, 0),
                       (48, 16, 8)]:
        x = torch.arange(Tp, dtype=torch.float).view(Tp, 1, 1).repeat(1, 2, 1)
        pad = torch.zeros(2, Tp, dtype=torch.bool)
        pad[1, Tp - 1] = True
        xo, po, am = ref.gen_block_attn_mask(x, pad, m, r)
        out[f"blk_{Tp}_{m}_{r}_src"] = xo[:, 0, 0].numpy().astype(np.int32)  # which frame each row copies
        out[f"blk_{Tp}_{m}_{r}_pad"] = po.numpy()
        out[f"blk_{Tp}_{m}_{r}_mask"] = np.packbits((am != 0).numpy(), axis=1)
        vals = torch.unique(am)
        out[f"blk_{Tp}_{m}_{r}_vals"] = vals.numpy()
    # G4: sinusoid rows (fs/modules/sinusoidal_positional_embedding.py:35-58)
    emb = ref.SinusoidalPositionalEmbedding(768, padding_idx=1, init_size=8002)
    out["sin768_rows"] = emb.weights[[0, 1, 2, 3, 500, 8001]].numpy()
    pe = emb(torch.tensor([[False, False, True, False], [False, False, False, False]]))
    out["sin768_bool_input"] = pe[:, :, :4].numpy()
    np.savez_compressed(os.path.join(HERE, "host_rng.npz"), **out)
    print("host_rng.npz", len(out), "arrays")


def run_model(tag, cfg_over, B, L, train, seed, record_layerdrop=False):
    cfg = ref_import.make_cfg(ref, **{k: v for k, v in cfg_over.items() if not k.startswith("_")})
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    model = ref.Wav2VecSModel(cfg)
    # perturb the LN/bias params so their gradients/paths are not trivially symmetric
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias") or "layer_norm" in n or ".2.1." in n or ".2.weight" in n:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
    model.train(train)
    source = torch.randn(B, L, generator=g)
    rec = {}

    # --- record the host draws the reference makes --------------------------------
    orig_cmi = w2.compute_mask_indices

    def cmi(*a, **k):
        m = orig_cmi(*a, **k)
        rec["mask_indices"] = m.copy()
        return m

    orig_sn = model.sample_negatives

    def sn(y, num):
        negs, idx = orig_sn(y, num)
        rec["neg_idx"] = idx.clone()
        return negs, idx

    orig_gs = F.gumbel_softmax

    def gs(logits, tau=1, hard=False, eps=1e-10, dim=-1):
        st = torch.get_rng_state()
        out = orig_gs(logits, tau=tau, hard=hard, eps=eps, dim=dim)
        after = torch.get_rng_state()
        torch.set_rng_state(st)
        rec["gumbel_noise"] = -torch.empty_like(logits, memory_format=torch.legacy_contiguous_format).exponential_().log()
        torch.set_rng_state(after)
        rec["tau"] = float(tau)
        return out

    orig_rr = np.random.random
    draws = []

    def rr(*a, **k):
        v = orig_rr(*a, **k)
        draws.append(v)
        return v

    orig_ri = random.randint
    ctx = []

    def ri(a, b):
        v = orig_ri(a, b)
        ctx.append(v)
        return v

    inter = {}
    hooks = []

    def keep(name):
        def h(mod, inp, out):
            inter[name] = (out[0] if isinstance(out, tuple) else out).detach().clone()
        return h

    hooks.append(model.feature_extractor.conv_layers[0].register_forward_hook(keep("conv0")))
    hooks.append(model.feature_extractor.register_forward_hook(keep("conv_out")))
    hooks.append(model.layer_norm.register_forward_hook(keep("features")))
    hooks.append(model.encoder.layers[0].register_forward_hook(keep("layer0")))
    hooks.append(model.encoder.register_forward_hook(keep("enc_out")))
    hooks.append(model.project_q.register_forward_hook(keep("yq")))
    hooks.append(model.final_proj.register_forward_hook(keep("xf")))
    hooks.append(model.quantizer.register_forward_hook(lambda m, i, o: inter.__setitem__("q", o["x"].detach().clone())))

    w2.compute_mask_indices = cmi
    model.sample_negatives = sn
    F.gumbel_softmax = gs
    np.random.random = rr
    random.randint = ri
    try:
        model.set_num_updates(1000)
        net = model(source)
    finally:
        w2.compute_mask_indices = orig_cmi
        F.gumbel_softmax = orig_gs
        np.random.random = orig_rr
        random.randint = orig_ri
        for h in hooks:
            h.remove()
    # --- criterion arithmetic (fs/criterions/wav2vec_criterion.py:64-107), via the model hooks
    logits = model.get_logits(net).float()
    target = model.get_targets(None, net)
    loss = F.cross_entropy(logits, target, reduction="sum")
    sample_size = target.numel()
    weights = cfg_over.get("_loss_weights", [0.1, 10.0])
    extra = model.get_extra_losses(net)
    loss0 = loss.detach().clone()
    for p, coef in zip(extra, weights):
        if coef != 0 and p is not None:
            loss = loss + coef * p.float() * sample_size
    model.zero_grad()
    if train:
        loss.backward()
    out = {"source": source.numpy(), "loss": np.array([loss.item()]), "loss0": np.array([loss0.item()]),
           "sample_size": np.array([sample_size]), "logits": logits.detach().numpy(),
           "features_pen": np.array([net["features_pen"].item()]),
           "prob_perplexity": np.array([net["prob_perplexity"].item()]),
           "code_perplexity": np.array([net["code_perplexity"].item()]),
           "temp": np.array([float(net["temp"])]), "loss_weights": np.array(weights),
           "mask_indices": rec["mask_indices"], "neg_idx": rec["neg_idx"].numpy()}
    if "gumbel_noise" in rec:
        out["gumbel_noise"] = rec["gumbel_noise"].numpy()
        out["tau"] = np.array([rec["tau"]])
    if record_layerdrop:
        out["layerdrop_draws"] = np.array(draws)
        out["context_draws"] = np.array(ctx)
    for k, v in inter.items():
        out["act." + k] = v.numpy()
    for n, p in model.state_dict().items():
        out["param." + n] = p.numpy()
    for n, p in (model.named_parameters() if train else []):
        out["grad." + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
        out["hasgrad." + n] = np.array([p.grad is not None])
    cfg_keep = {k: v for k, v in cfg_over.items() if not k.startswith("_")}
    out["cfg_json"] = np.frombuffer(repr(cfg_keep).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
    print(tag, "loss", loss.item(), "sample_size", sample_size, "arrays", len(out))


TINY = dict(
    encoder_embed_dim=32, encoder_ffn_embed_dim=64, encoder_attention_heads=4, final_dim=16,
    latent_vars=8, latent_groups=2, num_negatives=10,
    conv_feature_layers="[(32, 10, 5)] + [(32, 3, 2)] * 4 + [(32,2,2)] * 2",
    dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, dropout_input=0.0,
    dropout_features=0.0, encoder_layerdrop=0.0, context_type="constant", main_context=8,
    right_context=4,
)

if __name__ == "__main__":
    gen_host()
    # G5a: base-style (post-LN, 12 layers -> layer_norm_num=1, no conv bias, grad mult 0.1), train mode
    run_model("tiny_base", dict(TINY, encoder_layers=12), B=2, L=16000, train=True, seed=1)
    # G5b: large-style branches: pre-LN, conv_bias, layer_norm_num=7 (encoder_layers != 12), odd T
    run_model("tiny_large", dict(TINY, encoder_layers=3, layer_norm_first=True, conv_bias=True,
                                 feature_grad_mult=1.0, _loss_weights=[0.1, 0.0]),
              B=3, L=16400, train=True, seed=2)
    # G5c: eval mode (hard one-hot quantizer), group-norm extractor ("default" mode)
    run_model("tiny_eval_gn", dict(TINY, encoder_layers=2, extractor_mode="default"), B=2, L=12000,
              train=False, seed=3)
    # G5d: LayerDrop + sampled context sizes recorded
    run_model("tiny_layerdrop", dict(TINY, encoder_layers=12, encoder_layerdrop=0.3,
                                     context_type="sampling"), B=2, L=16000, train=True, seed=4,
              record_layerdrop=True)
