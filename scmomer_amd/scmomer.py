"""Mirror of model/scmomer.py for the RNA-only configuration (``atac_config=None``): the frozen Performer RNA
encoder with its per-gene conv head, ``rna_projection`` 16906 -> projection_dim, L2 normalisation, ``rna_ln``, the
student encoder on the raw expression values and ``to_out`` on their concatenation (scmomer.py:96-130).
Same constructor, attribute names and ``state_dict`` keys as the reference, so its checkpoints load.

The encoder runs on the MI355X kernels; the tail is a handful of [B, 16906] x [16906, 128/256] library products
and 128/256-wide LayerNorms, left to torch.  The ATAC branch (a ViT) is not part of this path:
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
        rna_embeds = self.rna_ln(self._get_rna_features(rna_long))
        atac_embeds = self.encoder(rna_values)
        return self.to_out(torch.cat([atac_embeds, rna_embeds], 1))

    def _get_rna_features(self, rna_values=None):
        """scmomer.py:124-130"""
        rna_features = self.rna_projection(self.rna_model(rna_values))
        if self.config.normalize:
            rna_features = rna_features / rna_features.norm(dim=-1, keepdim=True)
        return rna_features
