"""Drop-in for the reference's ``modules/layers.py``: ``PatchTransformerEncoder``
and ``PixelWiseDotProduct`` with the same constructor arguments, forward
signatures and state_dict keys, computed by the hand-written HIP kernels.

``HipEncoderStack`` runs the four post-norm transformer layers of an
``nn.TransformerEncoder`` parameter holder through ``ocv_encoder_layer_fwd``.
The torch modules are kept only as parameter containers (identical key names
and initialisation to the reference); their own forward is never called.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import hip_ops


class HipEncoderStack:
    """Forward of ``nn.TransformerEncoder(layer, num_layers)`` (eval mode, dropout
    off) on batch-first [B, S, E] tokens.  With a key-padding mask the padded
    rows of the result are exact zeros, as torch's nested-tensor fast path
    returns them."""

    def __init__(self, encoder: nn.TransformerEncoder):
        self.encoder = encoder
        l0 = encoder.layers[0]
        self.n_heads = l0.self_attn.num_heads
        self.dim_ff = l0.linear1.out_features
        self.eps = l0.norm1.eps
        if l0.norm_first:
            raise NotImplementedError("pre-norm transformer layers are not part of the reference path")
        self._packed = [dict() for _ in encoder.layers]      # per layer: three-term-split weights, keyed on versions

    def __call__(self, x: torch.Tensor, key_padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.encoder.training:
            raise RuntimeError("the HIP path implements inference (eval mode) only")
        n = len(self.encoder.layers)
        zero = key_padding_mask is not None
        if hip_ops.token_split3_enabled() and self.dim_ff % 128 == 0:
            # default: three-term-split projections / feed-forward, the whole stack in 1 + 2 n launches
            params, keep = [], []
            for i, layer in enumerate(self.encoder.layers):
                st, k = hip_ops.layer_params(layer, self._packed[i])
                params.append(st)
                keep.append(k)
            out = hip_ops.encoder_stack(x, params, key_padding_mask, zero_padded_rows=zero, n_heads=self.n_heads,
                                        dim_ff=self.dim_ff, eps=self.eps)
            del keep
            return out
        cur = x                                              # OCV_TOKENS=fp32: exact-fp32 kernels, four launches per layer
        for i, layer in enumerate(self.encoder.layers):
            params, keep = hip_ops.layer_params(layer, None)
            cur = hip_ops.encoder_layer(cur, params, key_padding_mask, zero_padded_rows=(zero and i == n - 1),
                                        n_heads=self.n_heads, dim_ff=self.dim_ff, eps=self.eps)
            del keep
        return cur


class PatchTransformerEncoder(nn.Module):
    """reference modules/layers.py:5-24."""

    def __init__(self, in_channels, patch_size=10, embedding_dim=128, num_heads=4, max_seq_len=500):
        super().__init__()
        encoder_layers = nn.TransformerEncoderLayer(embedding_dim, num_heads, dim_feedforward=1024)
        self.transformer_encoder = nn.TransformerEncoder(encoder_layers, num_layers=4, enable_nested_tensor=False)
        self.embedding_convPxP = nn.Conv2d(in_channels, embedding_dim, kernel_size=patch_size, stride=patch_size, padding=0)
        self.positional_encodings = nn.Parameter(torch.rand(max_seq_len, embedding_dim), requires_grad=True)
        self._stack = HipEncoderStack(self.transformer_encoder)
        self._w_cl = hip_ops.ChannelsLastWeight()
        self._w_pe = hip_ops.PatchEmbedSplitWeight()

    def forward_batch_first(self, x: torch.Tensor) -> torch.Tensor:
        """B x S x E tokens (the layout the kernels work in)."""
        if self.embedding_convPxP.kernel_size != (16, 16):
            raise NotImplementedError("the patch-embedding kernel is built for 16x16 patches")
        S = (x.shape[2] // 16) * (x.shape[3] // 16)
        if S > self.positional_encodings.shape[0]:
            raise ValueError(f"sequence length {S} exceeds max_seq_len {self.positional_encodings.shape[0]}")
        tok = hip_ops.patch_embed_auto(x, self.embedding_convPxP.weight.detach(), self.embedding_convPxP.bias.detach(),
                                       self.positional_encodings.detach()[:S], self._w_cl, self._w_pe)
        return self._stack(tok)

    def forward(self, x):
        return self.forward_batch_first(x).permute(1, 0, 2)      # S, N, E as the reference returns it


class PixelWiseDotProduct(nn.Module):
    """reference modules/layers.py:27-36."""

    def forward(self, x, K):
        n, c, h, w = x.size()
        _, cout, ck = K.size()
        assert c == ck, "Number of channels in x and Embedding dimension (at dim 2) of K matrix must match"
        return hip_ops.pixel_dot(x, K)
