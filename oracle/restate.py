"""Functional fp32 CPU restatement of the reference hot path (ORACLE -- test
infrastructure, never imported by the product; see oracle/__init__.py).

Every function takes a flat ``state_dict`` (name -> tensor, reference key
names) plus plain Python configuration values and returns tensors.  The
transformer / attention / LayerNorm / softmax arithmetic that the HIP kernels
implement is written out explicitly; dense convolutions, batch-norm and
bilinear resize use the torch primitives (torch is present in the container
and is the reference's own backend for them).

Citations are relative to /root/reference.

Pinned by: tests/golden/*.npz (see tests/golden/make_golden.py).
Unpinned : ps_roi_align (torchvision 0.13.1 absent) -- restated from the
           published algorithm; EfficientNet-B5 encoder -- see effnet_ref.py.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

N_HEADS = 4          # modules/ObjCAViT.py:155,163 ; modules/layers.py:8
N_LAYERS = 4         # modules/ObjCAViT.py:156,161 ; modules/layers.py:9
LN_EPS = 1e-5        # torch nn.TransformerEncoderLayer default layer_norm_eps
PAD_VALUE = 1e-4     # modules/ObjCAViT.py:183,194


# ----------------------------------------------------------------------------
# a11: torch nn.TransformerEncoderLayer / nn.MultiheadAttention semantics
# ----------------------------------------------------------------------------
def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = LN_EPS) -> torch.Tensor:
    """LayerNorm over the last dim, biased variance (torch nn.LayerNorm)."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * w + b


def multi_head_attention(q_src: torch.Tensor, k_src: torch.Tensor, v_src: torch.Tensor,
                         in_w: torch.Tensor, in_b: torch.Tensor,
                         out_w: torch.Tensor, out_b: torch.Tensor,
                         key_padding_mask: Optional[torch.Tensor] = None,
                         n_heads: int = N_HEADS) -> torch.Tensor:
    """nn.MultiheadAttention(batch_first=True, need_weights=False) forward.

    q_src B x Sq x E, k_src / v_src B x Sk x E, key_padding_mask B x Sk bool
    (True = ignore).  Packed in_proj_weight 3E x E, scale 1/sqrt(E/heads),
    masked scores -> -inf, softmax over keys, out_proj E x E.
    (reference call sites modules/ObjCAViT.py:195-207; layer use :155-161)
    """
    B, Sq, E = q_src.shape
    Sk = k_src.shape[1]
    d = E // n_heads
    q = q_src @ in_w[:E].T + in_b[:E]
    k = k_src @ in_w[E:2 * E].T + in_b[E:2 * E]
    v = v_src @ in_w[2 * E:].T + in_b[2 * E:]
    q = q.view(B, Sq, n_heads, d).permute(0, 2, 1, 3)
    k = k.view(B, Sk, n_heads, d).permute(0, 2, 1, 3)
    v = v.view(B, Sk, n_heads, d).permute(0, 2, 1, 3)
    scores = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(d))        # B x H x Sq x Sk
    if key_padding_mask is not None:
        scores = scores.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    m = scores.max(dim=-1, keepdim=True).values
    p = torch.exp(scores - m)
    p = p / p.sum(dim=-1, keepdim=True)
    ctx = (p @ v).permute(0, 2, 1, 3).reshape(B, Sq, E)
    return ctx @ out_w.T + out_b


def encoder_layer(x: torch.Tensor, sd: SD, pfx: str,
                  key_padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One post-norm nn.TransformerEncoderLayer(d_model=128, nhead=4, ff=1024,
    relu, eval => dropout off): x = LN1(x + MHA(x)); x = LN2(x + W2 relu(W1 x)).
    """
    a = multi_head_attention(x, x, x,
                             sd[pfx + "self_attn.in_proj_weight"], sd[pfx + "self_attn.in_proj_bias"],
                             sd[pfx + "self_attn.out_proj.weight"], sd[pfx + "self_attn.out_proj.bias"],
                             key_padding_mask)
    x = layer_norm(x + a, sd[pfx + "norm1.weight"], sd[pfx + "norm1.bias"])
    h = torch.relu(x @ sd[pfx + "linear1.weight"].T + sd[pfx + "linear1.bias"])
    f = h @ sd[pfx + "linear2.weight"].T + sd[pfx + "linear2.bias"]
    return layer_norm(x + f, sd[pfx + "norm2.weight"], sd[pfx + "norm2.bias"])


def transformer_encoder(x: torch.Tensor, sd: SD, pfx: str,
                        key_padding_mask: Optional[torch.Tensor] = None,
                        n_layers: int = N_LAYERS) -> torch.Tensor:
    """nn.TransformerEncoder(layer, num_layers=4) in eval()+no_grad.

    With a key-padding mask and batch_first=True torch takes the nested-tensor
    fast path (torch/nn/modules/transformer.py:529-548 in torch 2.10) and
    returns padded rows as exact 0.0; valid rows only ever attend
    to valid keys.  x is B x S x E.
    """
    for i in range(n_layers):
        x = encoder_layer(x, sd, f"{pfx}layers.{i}.", key_padding_mask)
    if key_padding_mask is not None:
        x = x.masked_fill(key_padding_mask[..., None], 0.0)
    return x


# ----------------------------------------------------------------------------
# a9 / a10: modules/layers.py
# ----------------------------------------------------------------------------
def patch_transformer_encoder(x: torch.Tensor, sd: SD, pfx: str, patch: int = 16) -> torch.Tensor:
    """PatchTransformerEncoder.forward (modules/layers.py:16-24).  Returns
    S x B x E (seq-first).  Seq-first, no mask => same arithmetic per batch
    element as the batch-first layer."""
    emb = F.conv2d(x, sd[pfx + "embedding_convPxP.weight"], sd[pfx + "embedding_convPxP.bias"],
                   stride=patch).flatten(2)                                   # B x E x S  (:17)
    S = emb.shape[2]
    emb = emb + sd[pfx + "positional_encodings"][:S, :].T.unsqueeze(0)       # (:19)
    tok = emb.permute(0, 2, 1)                                                # B x S x E
    tok = transformer_encoder(tok, sd, pfx + "transformer_encoder.")          # (:23)
    return tok.permute(1, 0, 2)                                               # S x B x E


def pixel_wise_dot_product(x: torch.Tensor, K: torch.Tensor) -> torch.Tensor:
    """PixelWiseDotProduct.forward (modules/layers.py:31-36):
    y[n,q,h,w] = sum_c x[n,c,h,w] * K[n,q,c]."""
    n, c, h, w = x.shape
    _, cout, ck = K.shape
    assert c == ck, "Number of channels in x and Embedding dimension (at dim 2) of K matrix must match"
    y = torch.einsum("ncp,nqc->nqp", x.reshape(n, c, h * w), K)
    return y.reshape(n, cout, h, w)


def _mlp(x: torch.Tensor, sd: SD, pfx: str, idxs: Sequence[int], slope: float = 0.01) -> torch.Tensor:
    """nn.Sequential of Linear layers at positions ``idxs`` with LeakyReLU(0.01)
    between them."""
    for j, i in enumerate(idxs):
        x = x @ sd[f"{pfx}{i}.weight"].T + sd[f"{pfx}{i}.bias"]
        if j + 1 < len(idxs):
            x = F.leaky_relu(x, slope)
    return x


def bin_width_regressor(tok0: torch.Tensor, sd: SD, pfx: str, norm: str = "linear") -> torch.Tensor:
    """regressor + normalisation (modules/ObjCAViT.py:299-303,378-388;
    modules/miniViT.py:16-20,33-42)."""
    y = _mlp(tok0, sd, pfx, (0, 2, 4))
    if norm == "linear":
        y = torch.relu(y) + 0.1
    elif norm == "softmax":
        return torch.softmax(y, dim=1)
    else:
        y = torch.sigmoid(y)
    return y / y.sum(dim=1, keepdim=True)


# ----------------------------------------------------------------------------
# a8: modules/miniViT.py
# ----------------------------------------------------------------------------
def mvit_forward(x: torch.Tensor, sd: SD, pfx: str = "", n_query: int = 128,
                 patch: int = 16, norm: str = "linear") -> Tuple[torch.Tensor, torch.Tensor]:
    """mViT.forward (modules/miniViT.py:22-44) -> (bin_widths_normed B x 256,
    range_attention_maps B x 128 x h x w)."""
    tgt = patch_transformer_encoder(x, sd, pfx + "patch_transformer.", patch)          # S x B x E (:24)
    feat = F.conv2d(x, sd[pfx + "conv3x3.weight"], sd[pfx + "conv3x3.bias"], padding=1)  # (:25)
    head, queries = tgt[0], tgt[1:n_query + 1]                                         # (:27)
    queries = queries.permute(1, 0, 2)                                                 # (:30)
    ram = pixel_wise_dot_product(feat, queries)                                        # (:31)
    y = bin_width_regressor(head, sd, pfx + "regressor.", norm)
    return y, ram


# ----------------------------------------------------------------------------
# a7: GridRandomPositionalEmbeddings (modules/ObjCAViT.py:18-147)
# ----------------------------------------------------------------------------
def grid_sample_bilinear_zeros(inp: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    """F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=False)
    written out.  inp N x C x H x W, grid N x Ho x Wo x 2 (x, y in [-1, 1])."""
    N, C, H, W = inp.shape
    gx = ((grid[..., 0] + 1.0) * W - 1.0) / 2.0
    gy = ((grid[..., 1] + 1.0) * H - 1.0) / 2.0
    x0 = torch.floor(gx)
    y0 = torch.floor(gy)
    out = torch.zeros(N, C, grid.shape[1], grid.shape[2], dtype=inp.dtype)
    flat = inp.reshape(N, C, H * W)
    for dy in (0, 1):
        for dx in (0, 1):
            xi = x0 + dx
            yi = y0 + dy
            wgt = (1.0 - (gx - xi).abs()) * (1.0 - (gy - yi).abs())
            ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
            xi_c = xi.clamp(0, W - 1).long()
            yi_c = yi.clamp(0, H - 1).long()
            idx = (yi_c * W + xi_c).reshape(N, 1, -1).expand(N, C, -1)
            vals = torch.gather(flat, 2, idx).reshape(N, C, grid.shape[1], grid.shape[2])
            out = out + vals * (wgt * ok.to(inp.dtype)).unsqueeze(1)
    return out


def _roi_bilinear(g: torch.Tensor, y: float, x: float):
    """torchvision bilinear_interpolate at one (y, x) for all channels of a C x H x W grid (0 outside [-1, size])."""
    C, H, W = g.shape
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return None
    y = max(y, 0.0)
    x = max(x, 0.0)
    yl, xl = int(y), int(x)
    if yl >= H - 1:
        yh = yl = H - 1
        y = float(yl)
    else:
        yh = yl + 1
    if xl >= W - 1:
        xh = xl = W - 1
        x = float(xl)
    else:
        xh = xl + 1
    ly, lx = y - yl, x - xl
    hy, hx = 1.0 - ly, 1.0 - lx
    return hy * hx * g[:, yl, xl] + hy * lx * g[:, yl, xh] + ly * hx * g[:, yh, xl] + ly * lx * g[:, yh, xh]


def ps_roi_align_1x1(grid: torch.Tensor, boxes: torch.Tensor, spatial_scale: float) -> torch.Tensor:
    """torchvision.ops.ps_roi_align(input 1 x C x H x W, [boxes K x 4],
    output_size=[1,1], spatial_scale, sampling_ratio=-1) -> K x C.

    PARITY UNPINNED (torchvision==0.13.1, conda_environment_files/graphbins.yml:168,
    is absent from the container and the reference holds no fixture for it).
    Published algorithm (torchvision/csrc/ops/cpu/ps_roi_align_kernel.cpp,
    restated from memory of the public source): roi corners are scaled and
    shifted by -0.5 ("do not use rounding"); pooled 1x1 => bin = roi, with NO
    minimum extent (the 0.1 floor belongs to ps_roi_POOL); adaptive sampling
    grid ceil(roi_h) x ceil(roi_w); sample (iy, ix) at start + (i + .5) *
    extent / n; bilinear_interpolate returns 0 outside [-1, size], clamps
    negatives to 0 and the last row / column to itself; samples are summed and
    divided by the FULL count (a box without extent: 0 / 0 = NaN); with a 1x1
    output, output channel c reads input channel c.  Samples that cannot lie
    inside [-1, size] are skipped here by index window (same sum: they are 0),
    which bounds the loops by the grid size.
    Call sites: modules/ObjCAViT.py:128,144."""
    C, H, W = grid.shape[1:]
    g = grid[0].double()
    out = torch.zeros(boxes.shape[0], C, dtype=torch.float64)

    f32 = np.float32

    def window(start, step, n, size):
        start, step = float(start), float(step)
        lo = max(int(math.floor((-1.0 - start) / step - 0.5)) - 1, 0)
        hi = min(int(math.ceil((size - start) / step - 0.5)) + 1, n - 1)
        return lo, hi

    for n in range(boxes.shape[0]):
        # coordinate arithmetic in fp32 like the published kernel (T = float): ceil() and the inside test are
        # discontinuous, so the restatement must round where the reference rounds; the sum itself is kept in fp64
        x1, y1, x2, y2 = (f32(float(v)) * f32(spatial_scale) - f32(0.5) for v in boxes[n])
        rw, rh = x2 - x1, y2 - y1
        if not all(math.isfinite(float(v)) for v in (x1, y1, rw, rh)):
            out[n] = float("nan")
            continue
        gh, gw = max(int(math.ceil(float(rh))), 0), max(int(math.ceil(float(rw))), 0)
        if gh * gw == 0:
            out[n] = float("nan")                      # 0 / 0
            continue
        acc = torch.zeros(C, dtype=torch.float64)
        ylo, yhi = window(y1, rh / gh, gh, H)
        xlo, xhi = window(x1, rw / gw, gw, W)
        for iy in range(ylo, yhi + 1):
            yy = y1 + f32(iy + 0.5) * rh / f32(gh)
            for ix in range(xlo, xhi + 1):
                xx = x1 + f32(ix + 0.5) * rw / f32(gw)
                v = _roi_bilinear(g, float(yy), float(xx))
                if v is not None:
                    acc += v
        out[n] = acc / (gh * gw)
    return out.to(grid.dtype)


def _xywh_to_xyxy_clamped(c: torch.Tensor) -> torch.Tensor:
    """modules/ObjCAViT.py:115-124 / :134-143."""
    hw, hh = c[..., 2] / 2, c[..., 3] / 2
    return torch.stack([c[..., 0] - hw, c[..., 1] - hh, c[..., 0] + hw, c[..., 1] + hh], dim=-1).clamp(min=0.0)


def grid_random_pos_emb(table: torch.Tensor, coords: torch.Tensor, feat_hw: Tuple[int, int],
                        patch: int, mode: str, space: str, factor: float = 2.0) -> torch.Tensor:
    """GridRandomPositionalEmbeddings.forward (modules/ObjCAViT.py:50-147),
    reproduced literally including the normalisation quirks.

    table: positional_encodings (L x E).  coords: N x {2,4} ("obj") or
    B x S x {2,4} ("img").  Returns N x E or B x S x E."""
    fh, fw = feat_hw
    gh, gw = math.ceil(fh / patch), math.ceil(fw / patch)                          # :78-79
    E = table.shape[1]
    grid = table[:gh * gw].view(gh, gw, E).permute(2, 0, 1).unsqueeze(0).contiguous()   # :82-83
    if mode == "centre":
        nc = coords.clone()
        if space == "img":
            # indexes dim 1 of a B x S x 2 tensor: TOKENS 0 and 1, not x / y  (:95-96)
            nc[:, 0] = ((nc[:, 0] / gh) * 2) - 1
            nc[:, 1] = ((nc[:, 1] / gw) * 2) - 1
            nc = nc.unsqueeze(1)                                                   # B x 1 x S x 2
            samples = grid_sample_bilinear_zeros(grid.expand(nc.shape[0], -1, -1, -1), nc)
            return samples.squeeze(2).permute(0, 2, 1).contiguous()                # B x S x E (:100)
        # "obj": x / image HEIGHT, y / image WIDTH  (:104-105)
        nc[:, 0] = ((nc[:, 0] / (fh * factor)) * 2) - 1
        nc[:, 1] = ((nc[:, 1] / (fw * factor)) * 2) - 1
        nc = nc.view(1, 1, nc.shape[0], 2)
        samples = grid_sample_bilinear_zeros(grid, nc)
        return samples.squeeze(2).squeeze(0).permute(1, 0).contiguous()            # N x E (:110)
    assert mode == "roi_align"
    if space == "img":
        xyxy = _xywh_to_xyxy_clamped(coords)                                       # B x S x 4
        return torch.stack([ps_roi_align_1x1(grid, b, 1.0 / patch) for b in xyxy], dim=0)   # :125-131
    xyxy = _xywh_to_xyxy_clamped(coords)
    return ps_roi_align_1x1(grid, xyxy, 1.0 / (patch * factor))                     # :144


# ----------------------------------------------------------------------------
# a6: SelfAttnCrossAttn.forward (modules/ObjCAViT.py:167-213)
# ----------------------------------------------------------------------------
def _pad_sequence(seqs: Sequence[torch.Tensor], value, n: Optional[int] = None) -> torch.Tensor:
    n = max(s.shape[0] for s in seqs) if n is None else n
    out = seqs[0].new_full((len(seqs), n) + tuple(seqs[0].shape[1:]), value)
    for i, s in enumerate(seqs):
        out[i, : s.shape[0]] = s
    return out


def saca_forward(img_tok: torch.Tensor, objs: Union[List[torch.Tensor], torch.Tensor], sd: SD, pfx: str,
                 no_obj_sa: bool = False, want_obj_out: bool = True, batch_nmax: Optional[int] = None
                 ) -> Tuple[torch.Tensor, Optional[torch.Tensor], dict]:
    """Returns (final_image_features B x S x E, final_object_features B x S x E,
    intermediates).  ``objs`` is a list of N_i x E tensors, or (saca_2, SURVEY
    Q3) a B x S x E tensor that is iterated over its batch dim.
    ``batch_nmax``: the images are a SLICE of a larger batch whose longest object
    list has that many entries -- pad_sequence (:180-183) would have padded to it."""
    B, S, E = img_tok.shape
    att_img = transformer_encoder(img_tok, sd, pfx + "image_transformer_encoder.")          # :169
    seqs = [o for o in objs]
    masks = _pad_sequence([torch.zeros(o.shape[0], dtype=torch.bool) for o in seqs], True, batch_nmax)  # :180-181
    feats = _pad_sequence(seqs, PAD_VALUE, batch_nmax)                                      # :183
    if no_obj_sa:
        att_obj = feats                                                                     # :186
    else:
        att_obj = transformer_encoder(feats, sd, pfx + "obj_transformer_encoder.", masks)   # :188
    amt = S - att_obj.shape[1]                                                              # :192
    assert amt >= 0, "more objects than image tokens"
    kpm = F.pad(masks, (0, amt), value=True)                                                # :193 (BACK)
    att_obj_p = F.pad(att_obj, (0, 0, amt, 0), value=PAD_VALUE)                             # :194 (FRONT)
    p1 = pfx + "cross_attn_obj_im."
    fin_img = multi_head_attention(att_img, att_obj_p, att_img,
                                   sd[p1 + "in_proj_weight"], sd[p1 + "in_proj_bias"],
                                   sd[p1 + "out_proj.weight"], sd[p1 + "out_proj.bias"], kpm)   # :195-201
    fin_obj = None
    if want_obj_out:
        p2 = pfx + "cross_attn_im_obj."
        fin_obj = multi_head_attention(att_obj_p, att_img, att_obj_p,
                                       sd[p2 + "in_proj_weight"], sd[p2 + "in_proj_bias"],
                                       sd[p2 + "out_proj.weight"], sd[p2 + "out_proj.bias"], None)  # :202-207
    inter = {"att_img": att_img, "att_obj": att_obj, "kpm": kpm}
    return fin_img, fin_obj, inter


# ----------------------------------------------------------------------------
# a4: ObjCAViT.forward (modules/ObjCAViT.py:306-390)
# ----------------------------------------------------------------------------
def patch_coords(B: int, gh: int, gw: int, patch: int = 16) -> torch.Tensor:
    """B x S x 4 = (x_centre, y_centre, patch, patch) in feature-map pixels
    (modules/ObjCAViT.py:336-347); S index = ph*gw + pw."""
    xs = torch.arange(gw).view(1, -1).expand(gh, -1)
    ys = torch.arange(gh).view(-1, 1).expand(-1, gw)
    pc = torch.stack([xs, ys], dim=0) * patch + patch // 2
    pc = pc.flatten(1).expand(B, -1, -1).permute(0, 2, 1).float()
    return torch.cat([pc, torch.ones_like(pc) * patch], dim=2)


_POS_MLP = (0, 2, 4, 6, 8)


def objcavit_forward(image_features: torch.Tensor, object_features: List[torch.Tensor],
                     object_xywh_list: List[Optional[torch.Tensor]], sd: SD, pfx: str = "", *,
                     strategy: str = "learned", no_obj_sa: bool = False, use_2_saca: bool = False,
                     n_query: int = 128, patch: int = 16, norm: str = "linear",
                     return_intermediates: bool = False, batch_nmax: Optional[int] = None):
    """-> (bin_widths_normed B x 256, range_attention_maps B x 128 x h x w).
    ``batch_nmax``: see saca_forward (these images as a slice of a larger batch)."""
    B, C, fh, fw = image_features.shape
    pe = pfx + "positional_encoder."
    objs = []
    for i, xywh in enumerate(object_xywh_list):                                            # :311
        if xywh is None:
            xywh = torch.zeros(1, 4) - 1                                                   # :313
        if strategy == "grid_random":
            pos = grid_random_pos_emb(sd[pe + "positional_encodings"], xywh[:, 0:2], (fh, fw), patch, "centre", "obj")
        elif strategy == "grid_random_roi_align":
            pos = grid_random_pos_emb(sd[pe + "positional_encodings"], xywh[:, 0:4], (fh, fw), patch, "roi_align", "obj")
        elif strategy == "learned_bbox_wh":
            pos = _mlp(xywh[:, 0:4], sd, pe, _POS_MLP)
        elif strategy == "learned":
            pos = _mlp(xywh[:, 0:2], sd, pe, _POS_MLP)
        else:
            raise SystemExit("Error: ObjCAViT positional embedding strategy not recognised.")   # :284
        emb = object_features[i] @ sd[pfx + "obj_embedding_layer.weight"].T + sd[pfx + "obj_embedding_layer.bias"]
        objs.append(emb + pos)                                                             # :330

    emb = F.conv2d(image_features, sd[pfx + "image_embedding_convPxP.weight"],
                   sd[pfx + "image_embedding_convPxP.bias"], stride=patch)                 # :333
    gh, gw = emb.shape[2], emb.shape[3]
    pc = patch_coords(B, gh, gw, patch)
    if strategy == "grid_random":
        ipos = grid_random_pos_emb(sd[pe + "positional_encodings"], pc[..., 0:2], (fh, fw), patch, "centre", "img")
    elif strategy == "grid_random_roi_align":
        ipos = grid_random_pos_emb(sd[pe + "positional_encodings"], pc[..., 0:4], (fh, fw), patch, "roi_align", "img")
    elif strategy == "learned_bbox_wh":
        ipos = _mlp(pc[..., 0:4], sd, pe, _POS_MLP)
    else:
        ipos = _mlp(pc[..., 0:2], sd, pe, _POS_MLP)
    tok = (emb.flatten(2) + ipos.permute(0, 2, 1)).permute(0, 2, 1)                        # :362-364

    inter = {"tokens_in": tok, "objs_in": objs}
    img, obj, i1 = saca_forward(tok, objs, sd, pfx + "saca_1.", no_obj_sa, want_obj_out=use_2_saca, batch_nmax=batch_nmax)   # :366
    inter["saca1_img"] = img
    inter["saca1_att_img"] = i1["att_img"]
    inter["saca1_att_obj"] = i1["att_obj"]
    if use_2_saca:
        inter["saca1_obj"] = obj
        img, obj, _ = saca_forward(img, obj, sd, pfx + "saca_2.", no_obj_sa, want_obj_out=False)      # :368
        inter["saca2_img"] = img

    head, queries = img[:, 0, :], img[:, 1:n_query + 1, :]                                 # :373
    feat = F.conv2d(image_features, sd[pfx + "conv3x3.weight"], sd[pfx + "conv3x3.bias"], padding=1)  # :374
    ram = pixel_wise_dot_product(feat, queries)                                            # :375
    y = bin_width_regressor(head, sd, pfx + "regressor.", norm)                            # :378-388
    if return_intermediates:
        inter["feat"] = feat
        inter["queries"] = queries
        return y, ram, inter
    return y, ram


# ----------------------------------------------------------------------------
# a1 / a2: bin head glue (modules/GraphBins.py:109-119 == modules/AdaBins.py:77-87)
# ----------------------------------------------------------------------------
def bin_head(bin_widths_normed: torch.Tensor, ram: torch.Tensor, conv_w: torch.Tensor, conv_b: torch.Tensor,
             min_depth: float, max_depth: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (depth_pred B x 1 x h x w, bin_edges B x 257)."""
    logits = F.conv2d(ram, conv_w, conv_b)                          # conv_out[0]
    m = logits.max(dim=1, keepdim=True).values
    p = torch.exp(logits - m)
    p = p / p.sum(dim=1, keepdim=True)                              # Softmax(dim=1)
    widths = (max_depth - min_depth) * bin_widths_normed            # GraphBins.py:111
    widths = F.pad(widths, (1, 0), mode="constant", value=min_depth)  # :112
    edges = torch.cumsum(widths, dim=1)                             # :113
    centers = 0.5 * (edges[:, :-1] + edges[:, 1:])                  # :115
    depth = torch.sum(p * centers.view(*centers.shape, 1, 1), dim=1, keepdim=True)   # :119
    return depth, edges


# ----------------------------------------------------------------------------
# a3: DenseFeatureExtractor decoder (modules/DenseFeatureExtractor.py:30-118)
# ----------------------------------------------------------------------------
def _bn_eval(x: torch.Tensor, sd: SD, pfx: str, eps: float = 1e-5) -> torch.Tensor:
    return F.batch_norm(x, sd[pfx + "running_mean"], sd[pfx + "running_var"],
                        sd[pfx + "weight"], sd[pfx + "bias"], False, 0.0, eps)


def upsample_with_skip(x: torch.Tensor, skip: torch.Tensor, sd: SD, pfx: str) -> torch.Tensor:
    """UpSampleWithSkip.forward (:44-47) with _net = conv,bn,lrelu,conv,bn,lrelu (:37-42)."""
    up = F.interpolate(x, size=[skip.size(2), skip.size(3)], mode="bilinear", align_corners=True)
    f = torch.cat([up, skip], dim=1)
    f = F.leaky_relu(_bn_eval(F.conv2d(f, sd[pfx + "_net.0.weight"], sd[pfx + "_net.0.bias"], padding=1), sd, pfx + "_net.1."), 0.01)
    f = F.leaky_relu(_bn_eval(F.conv2d(f, sd[pfx + "_net.3.weight"], sd[pfx + "_net.3.bias"], padding=1), sd, pfx + "_net.4."), 0.01)
    return f


def decoder_forward(features: Sequence[torch.Tensor], sd: SD, pfx: str = "",
                    feature_select: Sequence[int] = (4, 5, 6, 8, 11), do_final_upscale: bool = False) -> torch.Tensor:
    """Decoder.forward (:104-118).  conv2 is a 1x1 conv WITH padding=1 (:57)."""
    b0, b1, b2, b3, b4 = [features[i] for i in feature_select]
    x = F.conv2d(b4, sd[pfx + "conv2.weight"], sd[pfx + "conv2.bias"], padding=1)
    x = upsample_with_skip(x, b3, sd, pfx + "up1.")
    x = upsample_with_skip(x, b2, sd, pfx + "up2.")
    x = upsample_with_skip(x, b1, sd, pfx + "up3.")
    x = upsample_with_skip(x, b0, sd, pfx + "up4.")
    if do_final_upscale:
        x = upsample_with_skip(x, features[0], sd, pfx + "final_upscale.")
    return F.conv2d(x, sd[pfx + "conv3.weight"], sd[pfx + "conv3.bias"], padding=1)


# ----------------------------------------------------------------------------
# Boundary A: full models
# ----------------------------------------------------------------------------
def dense_features(image: torch.Tensor, sd: SD, pfx: str = "dense_feature_extractor.", do_final_upscale: bool = False) -> torch.Tensor:
    """DenseFeatureExtractor.forward (:195-198) with the EfficientNet-B5 encoder
    restated in effnet_ref.py (encoder arithmetic: parity unpinned)."""
    from .effnet_ref import encoder_features
    feats = encoder_features(image, sd, pfx + "encoder.original_model.")
    return decoder_forward(feats, sd, pfx + "decoder.", do_final_upscale=do_final_upscale)   # :99-101,116-117


def adabins_forward(image: torch.Tensor, sd: SD, min_depth: float, max_depth: float, do_final_upscale: bool = False):
    """AdaBins.forward (modules/AdaBins.py:73-89) -> (depth_pred, bin_edges).  ``do_final_upscale``
    (modules/AdaBins.py:43, DenseFeatureExtractor.py:99-101,116-117): the decoder's fifth stage, features at full resolution."""
    unet = dense_features(image, sd, do_final_upscale=do_final_upscale)
    y, ram = mvit_forward(unet, sd, "adaptive_bins_layer.")
    return bin_head(y, ram, sd["conv_out.0.weight"], sd["conv_out.0.bias"], min_depth, max_depth)


def graphbins_forward(image: torch.Tensor, object_features: List[torch.Tensor],
                      object_xywh_list: List[Optional[torch.Tensor]], sd: SD,
                      min_depth: float, max_depth: float, do_final_upscale: bool = False, **objcavit_kw):
    """GraphBins.forward (modules/GraphBins.py:81-121) with the frozen detector /
    language producers replaced by their outputs (object_features list of
    N_i x 512, object_xywh_list) -> (depth_pred, bin_edges)."""
    feats = dense_features(image, sd, do_final_upscale=do_final_upscale)
    y, ram = objcavit_forward(feats, [f.float() for f in object_features], object_xywh_list, sd,
                              "objcavit.", **objcavit_kw)
    return bin_head(y, ram, sd["conv_out.0.weight"], sd["conv_out.0.bias"], min_depth, max_depth)


# ----------------------------------------------------------------------------
# parity metric (metrics/AbsRel.py:23)
# ----------------------------------------------------------------------------
def abs_rel(pred: torch.Tensor, ref: torch.Tensor) -> float:
    return float(torch.mean(torch.abs(ref - pred) / ref))
