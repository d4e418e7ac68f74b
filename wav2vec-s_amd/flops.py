"""Algorithmic FLOPs of one pre-training step, by SURVEY.md section 8(d)'s calculator (MAC = 2; backward = 2 x forward,
step = 3 x forward): conv 2*Cin*Cout*k*L_out; linears per token 2*4*d^2 + 2*2*d*ffn; attention counted on the
UNMASKED (query, key) pairs only, 4*pairs*d per layer; quantizer / project_q / final_proj / InfoNCE products.
This is the reference algorithm's arithmetic - dead-code eliminations of this build (rows of the last layer nobody reads)
do not lower it, LayerDrop-ped layers and the sampled block context do (they change what the reference computes too).
Host arithmetic only; used by bench.py for `roofline.step`."""


def attention_pairs(Tp: int, m: int, r: int) -> int:
    """Visible (query, key) pairs of gen_block_attn_mask (fs/models/wav2vec/wav2vec_S.py:444-489): a main query of block b
    sees the main keys of blocks <= b and the r right-context copies of its own block; each of those r copies sees the
    same keys."""
    nblk_rc = Tp // m if r > 0 else 0
    pairs = 0
    b = 0
    t = 0
    while t < Tp:
        mb = min(m, Tp - t)
        rb = r if b < nblk_rc else 0
        keys = min((b + 1) * m, Tp) + rb
        pairs += (mb + rb) * keys
        t += mb
        b += 1
    return pairs


def forward_flops(cfg, B: int, L: int, *, m: int, r: int, layers_kept: int, M: int) -> dict:
    convs = cfg.conv_layers
    out = {}
    Lc, cin, conv = L, 1, 0
    for (c, k, s) in convs:
        Lc = (Lc - k) // s + 1
        conv += 2 * cin * c * k * Lc
        cin = c
    T = Lc
    C0 = convs[-1][0]
    E, Fd, G, V = cfg.encoder_embed_dim, cfg.encoder_ffn_embed_dim, cfg.latent_groups, cfg.latent_vars
    Tp = T + ((-T) % cfg.required_seq_len_multiple)
    N = Tp + (Tp // m) * r
    final = cfg.final_dim if cfg.final_dim > 0 else E
    vq = cfg.latent_dim if cfg.latent_dim > 0 else final
    out["conv"] = B * conv
    out["post_proj"] = B * T * 2 * C0 * E if C0 != E else 0
    out["encoder_linears"] = B * N * layers_kept * (2 * 4 * E * E + 2 * 2 * E * Fd)
    out["attention"] = B * layers_kept * 4 * attention_pairs(Tp, m, r) * E
    out["quantizer"] = B * M * (2 * C0 * G * V + 2 * G * V * (vq // G))
    out["project_q"] = B * M * 2 * vq * final
    out["final_proj"] = B * M * 2 * E * final
    out["infonce"] = B * M * (cfg.num_negatives + 1) * 2 * final
    out["total"] = sum(out.values())
    out["T"], out["N"] = T, N
    return out


def step_flops_from_state(st) -> float:
    """3 x forward FLOPs of the step whose engine.State is ``st`` (its own sampled context and LayerDrop outcome)."""
    f = forward_flops(st.cfg, st.B, st.source.shape[1], m=st.m, r=st.r, layers_kept=len(st.kept), M=st.M)
    return 3.0 * f["total"]


def _main_query_pairs(Tp: int, m: int, r: int) -> int:
    """``attention_pairs`` without the right-context copies as QUERIES (their outputs are dropped, wav2vec_S.py:425-427)."""
    nblk_rc = Tp // m if r > 0 else 0
    pairs, b, t = 0, 0, 0
    while t < Tp:
        mb = min(m, Tp - t)
        pairs += mb * (min((b + 1) * m, Tp) + (r if b < nblk_rc else 0))
        t += mb
        b += 1
    return pairs


def pruned_forward_flops(st) -> float:
    """Forward FLOPs of the reference's arithmetic that this build does NOT execute in the step of ``st``: of the last kept
    encoder layer only the masked frames are read downstream (fs/models/wav2vec/wav2vec2.py:590, 641), so its attention runs
    for the main-frame queries only and out_proj / fc1 / fc2 on the B*M selected rows (engine.forward, selected-rows mode)."""
    if not getattr(st, "enc_is_sel", False) or not st.kept:
        return 0.0
    cfg = st.cfg
    E, Fd = cfg.encoder_embed_dim, cfg.encoder_ffn_embed_dim
    rows_all, rows_sel = st.B * st.N, st.B * st.M
    lin = (rows_all - rows_sel) * (2 * E * E + 2 * 2 * E * Fd)
    att = st.B * 4 * (attention_pairs(st.Tp, st.m, st.r) - _main_query_pairs(st.Tp, st.m, st.r)) * E
    return float(lin + att)


def executed_step_flops_from_state(st) -> float:
    """``step_flops_from_state`` net of the pruned rows (3 x: forward, data gradient, weight gradient are all skipped)."""
    return step_flops_from_state(st) - 3.0 * pruned_forward_flops(st)
