"""scmomer_amd: MI355X-native Performer RNA-encoder path of scMomer.

Host side mirrors the reference's module API (model/performer/performer_pytorch.py,
model/rna_encoder.py); all arithmetic runs in hand-written gfx950 kernels behind the C ABI of
libscmomer_hip.so.
"""
from . import _lib  # noqa: F401  (raises if the HIP library is missing: there is no fallback)
from .performer import (FastAttention, FeedForward, Gene2VecPositionalEmbedding, Performer, PerformerLM,
                        PreLayerNorm, SelfAttention, SequentialSequence, gaussian_orthogonal_random_matrix,
                        linear_attention, softmax_kernel)

from .rna_encoder import Identity  # noqa: E402,F401
from .scmomer import StudentEncoder, scMomer  # noqa: E402,F401

__all__ = ["Identity", "StudentEncoder", "scMomer", "FastAttention", "FeedForward", "Gene2VecPositionalEmbedding", "Performer", "PerformerLM",
           "PreLayerNorm", "SelfAttention", "SequentialSequence", "gaussian_orthogonal_random_matrix",
           "linear_attention", "softmax_kernel"]
