"""Wav2VecSConfig: the reference's model dataclass (fs/models/wav2vec/wav2vec_S.py:43-311) with
the same field names and defaults, so yaml/argparse overrides written for the reference apply
unchanged.  String-expression fields keep the reference convention (eval'd:
conv_feature_layers wav2vec2.py:309, latent_temp :362)."""
from dataclasses import dataclass, fields
from typing import List, Tuple

EXTRACTOR_MODE_CHOICES = ("default", "layer_norm")
MASKING_DISTRIBUTION_CHOICES = ("static", "uniform", "normal", "poisson")
LAYER_TYPE_CHOICES = ("transformer", "conformer")


@dataclass
class Wav2VecSConfig:
    extractor_mode: str = "default"
    encoder_layers: int = 12
    encoder_embed_dim: int = 768
    encoder_ffn_embed_dim: int = 3072
    encoder_attention_heads: int = 12
    activation_fn: str = "gelu"
    layer_type: str = "transformer"
    dropout: float = 0.1
    attention_dropout: float = 0.1
    activation_dropout: float = 0.0
    encoder_layerdrop: float = 0.0
    dropout_input: float = 0.0
    dropout_features: float = 0.0
    final_dim: int = 0
    layer_norm_first: bool = False
    conv_feature_layers: str = "[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512,2,2)] + [(512,2,2)]"
    conv_bias: bool = False
    logit_temp: float = 0.1
    quantize_targets: bool = False
    quantize_input: bool = False
    same_quantizer: bool = False
    target_glu: bool = False
    feature_grad_mult: float = 1.0
    quantizer_depth: int = 1
    quantizer_factor: int = 3
    latent_vars: int = 320
    latent_groups: int = 2
    latent_dim: int = 0
    mask_length: int = 10
    mask_prob: float = 0.65
    mask_selection: str = "static"
    mask_other: float = 0
    no_mask_overlap: bool = False
    mask_min_space: int = 1
    require_same_masks: bool = True
    mask_dropout: float = 0.0
    mask_channel_length: int = 10
    mask_channel_prob: float = 0.0
    mask_channel_selection: str = "static"
    mask_channel_other: float = 0
    no_mask_channel_overlap: bool = False
    mask_channel_min_space: int = 1
    num_negatives: int = 100
    negatives_from_everywhere: bool = False
    cross_sample_negatives: int = 0
    codebook_negatives: int = 0
    conv_pos: int = 128
    conv_pos_groups: int = 16
    pos_conv_depth: int = 1
    latent_temp: str = "(2, 0.5, 0.999995)"
    max_positions: int = 100000
    checkpoint_activations: bool = False
    required_seq_len_multiple: int = 2
    crop_seq_to_multiple: int = 1
    depthwise_conv_kernel_size: int = 31
    attn_type: str = ""
    pos_enc_type: str = "abs"
    fp16: bool = False
    context_type: str = "constant"
    main_context: int = 16
    right_context: int = 16
    load_pretrained_model_from: str = ""
    pos_type: str = "sin"

    # ---- derived ----
    @property
    def conv_layers(self) -> List[Tuple[int, int, int]]:
        return eval(self.conv_feature_layers)

    @property
    def layer_norm_num(self) -> int:
        return 1 if self.encoder_layers == 12 else 7  # wav2vec_S.py:325

    @property
    def latent_temp_tuple(self):
        t = self.latent_temp
        return tuple(eval(t)) if isinstance(t, str) else tuple(t)

    @classmethod
    def from_namespace(cls, args):
        """argparse.Namespace / any attribute bag -> config (missing fields keep defaults), the
        role base_architecture plays for Wav2Vec2Model.build_model (wav2vec2.py:422-429, 981-1048)."""
        kw = {}
        for f in fields(cls):
            if hasattr(args, f.name) and getattr(args, f.name) is not None:
                kw[f.name] = getattr(args, f.name)
        return cls(**kw)


def base_librispeech_config(**over) -> Wav2VecSConfig:
    """examples/wav2vec/config/pretraining/wav2vec-S_base_librispeech.yaml:50-77 model section."""
    kw = dict(quantize_targets=True, extractor_mode="layer_norm", final_dim=256, encoder_layerdrop=0.05,
              dropout_input=0.1, dropout_features=0.1, encoder_embed_dim=768, feature_grad_mult=0.1,
              main_context=16, right_context=8, context_type="sampling", pos_type="sin", mask_length=10,
              mask_prob=0.65, mask_selection="static",
              conv_feature_layers="[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512,2,2)] * 2")
    kw.update(over)
    return Wav2VecSConfig(**kw)


def large_librivox_config(**over) -> Wav2VecSConfig:
    """examples/wav2vec/config/pretraining/wav2vec-S_large_librivox.yaml:52-91 model section."""
    kw = dict(quantize_targets=True, extractor_mode="layer_norm", layer_norm_first=True, final_dim=768,
              latent_temp="[2.0,0.1,0.999995]", encoder_layerdrop=0.0, dropout_input=0.1, dropout_features=0.1,
              dropout=0.0, attention_dropout=0.1, conv_bias=True, encoder_layers=24, encoder_embed_dim=1024,
              encoder_ffn_embed_dim=4096, encoder_attention_heads=16, feature_grad_mult=1.0, max_positions=8000,
              main_context=16, right_context=8, context_type="sampling", pos_type="sin", mask_length=10,
              mask_prob=0.65, mask_selection="static",
              conv_feature_layers="[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512,2,2)] * 2")
    kw.update(over)
    return Wav2VecSConfig(**kw)
