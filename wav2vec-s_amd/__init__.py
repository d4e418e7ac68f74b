"""wav2vec-S pre-training hot path for AMD MI355X (gfx950): hand-written HIP kernels behind a
C ABI (include/w2vs.h), driven by a Python mirror of the fairseq wav2vec-S model API."""
__version__ = "0.1.0"

from .config import Wav2VecSConfig, base_librispeech_config, large_librivox_config  # noqa: E402,F401


def __getattr__(name):  # lazy: model/criterion import torch + the HIP library
    if name in ("Wav2VecSModel", "Wav2Vec2Model", "ConvFeatureExtractionModel", "TransformerEncoder",
                "TransformerSentenceEncoderLayer", "BlockwiseTransformerEncoder", "gen_block_attn_mask",
                "EXTRACTOR_MODE_CHOICES", "MASKING_DISTRIBUTION_CHOICES", "LAYER_TYPE_CHOICES"):
        from . import model
        return getattr(model, name)
    if name in ("BlockWiseWav2Vec2Model", "OnlineW2V2TransformerEncoder", "gen_block_atten_mask", "HipLinear"):
        from . import streaming      # row f1: rain/layers/unidirect_w2v2_encoder.py
        return getattr(streaming, name)
    if name in ("MHAJointNet", "TransformerJointerLayer", "ExpandMultiheadAttention"):
        from . import joiner         # row f4: rain/layers/attention_transducer.py:591-852
        return getattr(joiner, name)
    if name == "Wav2vecCriterion":
        from .criterion import Wav2vecCriterion
        return Wav2vecCriterion
    raise AttributeError(name)
