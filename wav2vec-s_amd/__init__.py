"""wav2vec-S pre-training hot path for AMD MI355X (gfx950): hand-written HIP kernels behind a
C ABI (include/w2vs.h), driven by a Python mirror of the fairseq wav2vec-S model API."""
__version__ = "0.1.0"
