"""Mirror of model/scmomer.py for the RNA-only configuration (``atac_config=None``): the frozen Performer RNA
encoder with its per-gene conv head, ``rna_projection`` 16906 -> projection_dim, L2 normalisation, ``rna_ln``, the
student encoder on the raw expression values and ``to_out`` on their concatenation (scmomer.py:96-130).
Same constructor, attribute names and ``state_dict`` keys as the reference, so its checkpoints load.

The encoder runs on the MI355X kernels.  The tail (two [B, 16906] x [16906, 128/256] products, normalisation, two
LayerNorms, GELU, two narrow products) is two launches of its own kernel where no gradient is wanted and dropout is
off (evaluation, extraction: ops.cell_tail); the S3 distillation step runs it, its decoders, losses and backward
on kernels of their own (s3.py, ops.DistillCall); any other use with gradients or in train mode goes through this
module's forward, where the tail is torch library code.  The ATAC branch (a ViT) is not part of this path:
``atac_config`` must be None and mode='multimodal' raises.
"""
import torch
import torch.nn as nn

from . import ingest
from .rna_encoder import Performer


def exists(val):
    return val is not None


class StudentEncoder(nn.Module):
    """extract_cell_emb.py:27-40 (the same class is repeated in the reference's S3 / evaluation scripts)."""

    def __init__(self, input_dim=16906, output_dim=128, hidden_dim=256, dropout=0.1):
        super().__init__()
        self.encoder = nn.Sequential(
            nn.Linear(input_dim, hidden_dim),
            nn.LayerNorm(hidden_dim),
            nn.GELU(),
            nn.Dropout(dropout),
            nn.Linear(hidden_dim, output_dim),
        )

    def forward(self, rna_value):
        return self.encoder(rna_value)


class scMomer(nn.Module):
    """scmomer.py:9-39"""

    def __init__(self, config, atac_config, rna_decoder=None, atac_decoder=None, sub_task=None, encoder=None):
        super().__init__()
        if atac_config is not None:
            raise NotImplementedError("the ATAC encoder is outside the RNA-encoder path: pass atac_config=None")
        self.config = config
        self.rna_model = Performer()
        self.rna_projection = nn.Linear(16906, config.projection_dim, bias=False)
        self.to_out = nn.Linear(256, 128)
        self.RNA = rna_decoder
        self.ATAC = atac_decoder
        self.sub_task = sub_task
        self.encoder = encoder
        self.rna_ln = nn.LayerNorm(config.projection_dim)
        self.atac_ln = nn.LayerNorm(config.projection_dim)

    def forward(self, atac_values=None, rna_values=None, reconstruct=False, mode='multimodal'):
        """scmomer.py:41-94, mode='one' (RNA-only: the student encoder stands in for the ATAC embedding)."""
        if mode != 'one':
            if mode == 'multimodal':
                raise NotImplementedError("mode='multimodal' needs the ATAC encoder, which is outside this path")
            raise ValueError(f"Unknown mode: {mode}")
        rna_long = ingest.bin_cells(rna_values, ingest.RULE_SCMOMER, 5)   # round half to even, clamp to [0, 5]
        rna_embeds = self.rna_ln(self._get_rna_features(rna_long))
        atac_embeds = self.encoder(rna_values)
        out = self.to_out(torch.cat([atac_embeds, rna_embeds], 1))
        if reconstruct:
            return self.ATAC(out), self.RNA(out), atac_embeds
        return self.sub_task(out)

    def get_cell_embedding(self, rna_values):
        """scmomer.py:96-113 -> [batch, 128]"""
        rna_long = ingest.bin_cells(rna_values, ingest.RULE_SCMOMER, 5)   # round half to even, clamp to [0, 5]
        tail = self._tail_operands(rna_values)
        if tail is not None:      # evaluation / extraction: everything after the encoder in two launches
            from . import ops
            return ops.cell_tail(self.rna_model(rna_long).contiguous(), rna_values.contiguous(), *tail,
                                 normalize=bool(self.config.normalize))
        rna_embeds = self.rna_ln(self._get_rna_features(rna_long))
        atac_embeds = self.encoder(rna_values)
        return self.to_out(torch.cat([atac_embeds, rna_embeds], 1))

    def _tail_operands(self, rna_values):
        """The weights of the tail in the order ops.cell_tail takes them, or None when the fused tail does not apply:
        gradients wanted, dropout active (train mode), another encoder than the student MLP, or tensors that are
        not f32 on the GPU."""
        from . import ops
        enc = getattr(self.encoder, "encoder", None)
        if torch.is_grad_enabled() or self.training or not isinstance(self.encoder, StudentEncoder) or enc is None:
            return None
        if not (rna_values.is_cuda and rna_values.dtype == torch.float32 and rna_values.dim() == 2):
            return None
        fc1, ln, fc2 = enc[0], enc[1], enc[4]
        ws = (self.rna_projection.weight, (self.rna_ln.weight, self.rna_ln.bias), fc1.weight, fc1.bias,
              (ln.weight, ln.bias), fc2.weight, fc2.bias, self.to_out.weight, self.to_out.bias)
        flat = [t for w in ws for t in (w if isinstance(w, tuple) else (w,))]
        if not all(t is not None and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in flat):
            return None
        p, hd, o = self.rna_projection.weight.shape[0], fc1.weight.shape[0], self.to_out.weight.shape[0]
        if not ops.cell_tail_supported(p, hd, o) or tuple(fc2.weight.shape) != (p, hd) or \
                self.to_out.weight.shape[1] != 2 * p or self.rna_ln.eps != 1e-5 or ln.eps != 1e-5:
            return None
        return tuple(w if isinstance(w, tuple) else w.detach() for w in ws)

    def _get_rna_features(self, rna_values=None):
        """scmomer.py:124-130"""
        rna_features = self.rna_projection(self.rna_model(rna_values))
        if self.config.normalize:
            rna_features = rna_features / rna_features.norm(dim=-1, keepdim=True)
        return rna_features
