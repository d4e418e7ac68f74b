"""ctypes binding of libw2vs.so (the C ABI in include/w2vs.h).

There is NO fallback: if the HIP library is missing or a call is rejected, this raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("W2VS_LIB", os.path.join(_HERE, "libw2vs.so"))   # W2VS_LIB: experiment builds only

vp, i32, i64, u64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float


class GemmDesc(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("C2", vp), ("Cf", vp), ("bias", vp), ("aux", vp),
                ("M", i32), ("N", i32), ("K", i32), ("batch", i32),
                ("lda", i64), ("ldb", i64), ("ldc", i64), ("a_off", i64),
                ("sA", i64), ("sB", i64), ("sC", i64), ("a_bytes", i64), ("b_bytes", i64), ("c_elems", i64),
                ("epi", i32), ("alpha", f32), ("colsum", vp), ("ws", vp), ("ws_bytes", i64), ("zk_col", i32), ("zk_k", i32), ("overwrite", i32)]


class LnFwdDesc(C.Structure):
    _fields_ = [("x", vp), ("res", vp), ("gamma", vp), ("beta", vp), ("y", vp), ("sum_out", vp),
                ("mean", vp), ("rstd", vp), ("sumsq", vp), ("rows", i64), ("C", i32), ("gelu", i32),
                ("p_drop", f32), ("seed", u64)]


class LnBwdDesc(C.Structure):
    _fields_ = [("x", vp), ("gamma", vp), ("beta", vp), ("mean", vp), ("rstd", vp), ("dy", vp), ("dsum", vp),
                ("aux", vp), ("dx", vp), ("dres", vp), ("dgamma", vp), ("dbeta", vp), ("rows", i64),
                ("C", i32), ("gelu", i32), ("p_drop", f32), ("seed", u64), ("out_scale", f32), ("pen_coef", f32),
                ("pen_coef_dev", vp), ("ws", vp), ("ws_bytes", i64)]


class EncPrologueDesc(C.Structure):
    _fields_ = [("x", vp), ("mask", vp), ("pad", vp), ("pos", vp), ("mask_emb", vp), ("pos_table", vp),
                ("gamma", vp), ("beta", vp), ("out", vp), ("mean", vp), ("rstd", vp), ("src", vp),
                ("dout", vp), ("dx", vp), ("dmask_emb", vp), ("dgamma", vp), ("dbeta", vp),
                ("copy_start", vp), ("copy_list", vp),
                ("p_in", f32), ("p_enc", f32), ("seed_in", u64), ("seed_enc", u64),
                ("apply_ln", i32), ("B", i32), ("T", i32), ("Tp", i32), ("N", i32), ("C", i32)]


class AttnDesc(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("v", vp), ("o", vp), ("lse", vp), ("kpad", vp),
                ("dout", vp), ("delta", vp), ("dq", vp), ("dk", vp), ("dv", vp),
                ("ld", i64), ("ldo", i64), ("sb", i64), ("sbo", i64),
                ("B", i32), ("H", i32), ("N", i32), ("Tp", i32), ("m", i32), ("r", i32), ("head_dim", i32),
                ("scale", f32), ("p_drop", f32), ("seed", u64), ("Nq", i32), ("mq", i32), ("ldq", i64), ("sbq", i64), ("drop_bits", vp)]


class QuantDesc(C.Structure):
    _fields_ = [("logits", vp), ("noise", vp), ("vars", vp), ("q", vp), ("idx", vp), ("hard_cnt", vp),
                ("prob_sum", vp), ("ppl_out", vp), ("cvec_out", vp), ("dq", vp), ("dsoft", vp), ("cvec", vp),
                ("dlogits", vp), ("dvars", vp), ("ppl_grad", f32), ("tau", f32),
                ("R", i32), ("G", i32), ("V", i32), ("D", i32), ("training", i32), ("seed", u64), ("ppl_grad_dev", vp),
                ("logits_f32", vp), ("logit_bias", vp), ("dsoft_f32", vp)]


class NceDesc(C.Structure):
    _fields_ = [("x", vp), ("y", vp), ("neg_idx", vp), ("logits", vp), ("xn", vp), ("yn", vp), ("dlogits", vp),
                ("dx", vp), ("dy", vp), ("dy_ws", vp),
                ("B", i32), ("M", i32), ("K", i32), ("C", i32), ("temp", f32), ("ws", vp), ("ws_bytes", i64)]


class TransposeItem(C.Structure):
    _fields_ = [("inp", vp), ("out", vp), ("R", i32), ("C", i32), ("ld_in", i64), ("ld_out", i64)]


class LayerDesc(C.Structure):
    _fields_ = ([(n, i32) for n in ("B", "N", "E", "F", "H", "Tp", "m", "r", "post_ln", "num_cu")]
                + [("p_drop", f32), ("p_attn", f32), ("seed_attn", u64), ("seed_drop1", u64), ("seed_drop2", u64)]
                + [(n, vp) for n in (
                    "kpad", "wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2", "ln2_g", "ln2_b",
                    "x_in", "qkv", "ctx", "lse", "s1", "mean1", "rstd1", "x1", "hpre", "h", "s2", "mean2", "rstd2",
                    "x_out", "tmp", "d_out", "d_in",
                    "g_wqkv", "g_bqkv", "g_wo", "g_bo", "g_ln1_g", "g_ln1_b", "g_w1", "g_b1", "g_w2", "g_b2",
                    "g_ln2_g", "g_ln2_b", "ws_e0", "ws_e1", "ws_e2", "ws_f", "ws_qkv", "wt_scratch", "delta",
                    "wqkv_t", "wo_t", "w1_t", "w2_t", "tn_ws")] + [("tn_ws_bytes", i64), ("sel_idx", vp), ("n_sel", i32),
                                                                            ("n_q", i32), ("ctx_sel", vp), ("xin_sel", vp), ("ws_e3", vp),
                                                                            ("stream_in", vp), ("d_stream_out", vp), ("d_stream_in", vp), ("drop_bits", vp),
                                                                            ("defer_wgrads", i32), ("wgrad_overwrite", i32), ("ws_s0", vp), ("ws_s1", vp), ("ln_part", vp), ("ln_part_bytes", i64)])


class CollateDesc(C.Structure):
    _fields_ = [("flat", vp), ("offset", vp), ("size", vp), ("crop_start", vp), ("out", vp), ("padding_mask", vp),
                ("partial", vp), ("B", i32), ("target", i32), ("width", i32), ("max_size", i32), ("normalize", i32), ("out_f32", i32)]


class InfonceLossDesc(C.Structure):
    _fields_ = [("logits", vp), ("R", i64), ("W", i32), ("pen_acc", vp), ("ppl", vp),
                ("w_ppl", f32), ("w_pen", f32), ("num_vars", f32), ("pen_norm", f32), ("sample_size", f32),
                ("loss", vp), ("vec", vp), ("dlogits", vp), ("scratch", vp)]


_DESCS = [GemmDesc, LnFwdDesc, LnBwdDesc, EncPrologueDesc, AttnDesc, QuantDesc, NceDesc, LayerDesc, CollateDesc, InfonceLossDesc]

EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GELU_SAVE, EPI_DGELU, EPI_F32, EPI_ADD, EPI_BIAS_GELU_SAVEG, EPI_MUL = range(9)

# name -> argtypes ; every entry returns int
_SIGS = {
    "w2vs_gemm_nt": [C.POINTER(GemmDesc), vp],
    "w2vs_gemm_tn": [C.POINTER(GemmDesc), i32, vp],
    "w2vs_gemm_tn_group": [vp, i32, i32, vp],
    "w2vs_prof_enable": [i32],
    "w2vs_prof_read": [i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i32)],
    "w2vs_prof_read_raw": [i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i32)],
    "w2vs_prof_launches": [i32],          # returns a count, not a status (read through load(), not call())
    "w2vs_prof_flops": [i32],             # returns a double (read through load(), restype set there)
    "w2vs_conv0_fwd": [vp] * 8 + [i32] * 5 + [vp],
    "w2vs_conv0_bwd": [vp] * 12 + [i32] * 5 + [vp],
    "w2vs_conv0_gn_fwd": [vp] * 7 + [i32] * 5 + [vp],
    "w2vs_conv0_gn_bwd": [vp] * 12 + [i32] * 5 + [vp],
    "w2vs_ln_fwd": [C.POINTER(LnFwdDesc), vp],
    "w2vs_ln_bwd": [C.POINTER(LnBwdDesc), vp],
    "w2vs_enc_prologue_fwd": [C.POINTER(EncPrologueDesc), vp],
    "w2vs_enc_prologue_bwd": [C.POINTER(EncPrologueDesc), vp],
    "w2vs_attn_fwd": [C.POINTER(AttnDesc), vp],
    "w2vs_attn_bwd": [C.POINTER(AttnDesc), vp],
    "w2vs_layer_fwd": [C.POINTER(LayerDesc), vp],
    "w2vs_layer_bwd": [C.POINTER(LayerDesc), vp],
    "w2vs_layer_wgrads": [vp, i32, vp],
    "w2vs_layer_wgrads_parts": [vp, vp, i32, vp],
    "w2vs_quant_fwd": [C.POINTER(QuantDesc), vp],
    "w2vs_quant_bwd": [C.POINTER(QuantDesc), vp],
    "w2vs_nce_fwd": [C.POINTER(NceDesc), vp],
    "w2vs_nce_bwd": [C.POINTER(NceDesc), vp],
    "w2vs_ce_rows": [vp, i64, i32, vp, vp, vp],
    "w2vs_infonce_loss": [C.POINTER(InfonceLossDesc), vp],
    "w2vs_infonce_loss_bwd": [vp, vp, i64, f32, f32, vp, vp],
    "w2vs_gather_rows": [vp, vp, vp, i64, i32, i32, vp],
    "w2vs_gemm_tune": [i32, i32, i32],
    "w2vs_gemm_tn8_max_split": [i32],
    "w2vs_gemm_last_group_form": [],
    "w2vs_attn_tune": [i32],
    "w2vs_attn_drop_bits_bytes": [i32, i32, i32, i32],     # returns a byte count (read through load())
    "w2vs_transpose2d": [vp, vp, i32, i32, i32, vp],
    "w2vs_transpose_multi": [vp, i32, vp],
    "w2vs_f32_to_bf16": [vp, vp, i64, f32, vp],
    "w2vs_bf16_to_f32": [vp, vp, i64, vp],
    "w2vs_colsum": [vp, vp, i64, i32, i64, vp],
    "w2vs_dropout": [vp, vp, i64, f32, u64, vp],
    "w2vs_relu_gate": [vp, vp, vp, i64, vp],
    "w2vs_adam_step": [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp, f32, vp],
    "w2vs_sumsq": [vp, i64, vp, vp],
    "w2vs_clip_scale": [vp, vp, f32, f32, vp, vp],
    "w2vs_clip_scale_acc": [vp, vp, f32, f32, vp, vp, vp],
    "w2vs_batch_by_size": [vp, i64, i64, i64, i32, vp, vp],
    "w2vs_collate_chunks": [i32],
    "w2vs_collate": [C.POINTER(CollateDesc), vp],
}
EXPORTS = ["w2vs_abi_version", "w2vs_last_error", "w2vs_sizeof"] + list(_SIGS)

_lib = None


class W2vsError(RuntimeError):
    pass


def load():
    """Load libw2vs.so; raise loudly if it is absent or does not match this binding."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise W2vsError(
            "libw2vs.so not found at %s - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C wav2vec-s_amd/csrc`). There is no CPU/eager fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.w2vs_last_error.restype = C.c_char_p
    lib.w2vs_sizeof.argtypes = [i32]
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_double if name == "w2vs_prof_flops" else C.c_int
    if lib.w2vs_abi_version() != 1:
        raise W2vsError("libw2vs ABI version mismatch")
    for i, d in enumerate(_DESCS):
        if lib.w2vs_sizeof(i) != C.sizeof(d):
            raise W2vsError("descriptor %s: C sizeof %d != binding %d" % (d.__name__, lib.w2vs_sizeof(i), C.sizeof(d)))
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise W2vsError("%s failed (%d): %s" % (name, rc, lib.w2vs_last_error().decode()))
