"""Mirror of model/rna_encoder.py: the scMomer RNA encoder = PerformerLM(7 tokens, dim 200,
depth 6, 10 heads, 16906 genes) whose ``to_out`` is replaced by a per-gene scalar head
(1 x 200 conv + ReLU), giving one value per gene: [B, N]."""
import torch
import torch.nn as nn

from .performer import PerformerLM


class Identity(torch.nn.Module):
    """rna_encoder.py:5-16.  Parameters keep the reference's names/shapes (conv1.weight [1,1,1,200],
    conv1.bias [1]); inside PerformerLM.forward the head runs fused with the final LayerNorm."""

    def __init__(self, dim: int = 200):
        super().__init__()
        self.conv1 = nn.Conv2d(1, 1, (1, dim))
        self.act = nn.ReLU()

    def forward(self, x):
        # stand-alone use on encodings [B,N,dim]: a dot product per token, ReLU
        w = self.conv1.weight.reshape(-1)
        return self.act(x @ w + self.conv1.bias)


class Performer(nn.Module):
    """rna_encoder.py:19-41"""

    def __init__(self, add_pooling_layer: bool = True, use_mask_token: bool = False, CLASS=7, SEQ_LEN=16906,
                 POS_EMBED_USING: bool = True):
        super().__init__()
        self.model = PerformerLM(num_tokens=CLASS, dim=200, depth=6, max_seq_len=SEQ_LEN, heads=10,
                                 local_attn_heads=0, g2v_position_emb=POS_EMBED_USING)
        self.model.to_out = Identity()

    def forward(self, x):
        return self.model(x)
