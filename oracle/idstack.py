"""Oracle (test infrastructure): CPU restatement of the reference's identity-conditioning modules (SURVEY.md section 8
row f-3) -- the reference's OWN code, so this file is PINNED: ``tests/golden/make_golden_idstack.py`` runs the real
classes of /root/reference/functions.py and /root/reference/attention.py and ``tests/test_oracle_golden.py`` replays the
stored vectors through this restatement (same parameter names, so one state_dict feeds both).

  ProjPlusModel            functions.py:490-522   faceid vector + CLIP hidden states -> 4 identity tokens
  FacePerceiverResampler   functions.py:454-488
  PerceiverAttention       functions.py:407-452   (latents attend to [x ; latents], scale d^-1/4 on q and on k)
  FeedForward              functions.py:390-397   (LayerNorm, Linear, GELU, Linear; no biases)
  AttentionMLP             functions.py:524-593   (1 learned latent, depth 8, dim 1024) -- FacialEncoder.visual_projection
  MLP / FuseModule         attention.py:50-69 / :10-48   (masked gather of the trigger-token rows, fuse, masked scatter)
  FacialEncoder            attention.py:72-88
  assemble_prompt_embeds   pipline_StableDiffusion_ConsistentID.py:176-209, :479-507 (the cat order the loop's .chunk(3) expects)
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn


def FeedForward(dim, mult=4):
    inner = int(dim * mult)
    return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, inner, bias=False), nn.GELU(), nn.Linear(inner, dim, bias=False))


def _heads(x, heads):
    b, n, w = x.shape
    return x.view(b, n, heads, w // heads).transpose(1, 2)


class PerceiverAttention(nn.Module):
    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        self.dim_head, self.heads = dim_head, heads
        inner = dim_head * heads
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)

    def forward(self, x, latents):
        x, latents = self.norm1(x), self.norm2(latents)
        b, l, _ = latents.shape
        q = _heads(self.to_q(latents), self.heads)
        k, v = self.to_kv(torch.cat((x, latents), dim=-2)).chunk(2, dim=-1)
        k, v = _heads(k, self.heads), _heads(v, self.heads)
        scale = 1 / math.sqrt(math.sqrt(self.dim_head))
        w = torch.softmax(((q * scale) @ (k * scale).transpose(-2, -1)).float(), dim=-1).type(q.dtype)
        out = (w @ v).permute(0, 2, 1, 3).reshape(b, l, -1)
        return self.to_out(out)


def _perceiver_layers(dim, depth, dim_head, heads, ff_mult):
    return nn.ModuleList([nn.ModuleList([PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads),
                                         FeedForward(dim=dim, mult=ff_mult)]) for _ in range(depth)])


class FacePerceiverResampler(nn.Module):
    def __init__(self, *, dim=768, depth=4, dim_head=64, heads=16, embedding_dim=1280, output_dim=768, ff_mult=4):
        super().__init__()
        self.proj_in = nn.Linear(embedding_dim, dim)
        self.proj_out = nn.Linear(dim, output_dim)
        self.norm_out = nn.LayerNorm(output_dim)
        self.layers = _perceiver_layers(dim, depth, dim_head, heads, ff_mult)

    def forward(self, latents, x):
        x = self.proj_in(x)
        for attn, ff in self.layers:
            latents = attn(x, latents) + latents
            latents = ff(latents) + latents
        return self.norm_out(self.proj_out(latents))


class ProjPlusModel(nn.Module):
    def __init__(self, cross_attention_dim=768, id_embeddings_dim=512, clip_embeddings_dim=1280, num_tokens=4):
        super().__init__()
        self.cross_attention_dim, self.num_tokens = cross_attention_dim, num_tokens
        self.proj = nn.Sequential(nn.Linear(id_embeddings_dim, id_embeddings_dim * 2), nn.GELU(),
                                  nn.Linear(id_embeddings_dim * 2, cross_attention_dim * num_tokens))
        self.norm = nn.LayerNorm(cross_attention_dim)
        self.perceiver_resampler = FacePerceiverResampler(
            dim=cross_attention_dim, depth=4, dim_head=64, heads=cross_attention_dim // 64,
            embedding_dim=clip_embeddings_dim, output_dim=cross_attention_dim, ff_mult=4)

    def forward(self, id_embeds, clip_embeds, shortcut=False, scale=1.0):
        x = self.norm(self.proj(id_embeds).reshape(-1, self.num_tokens, self.cross_attention_dim))
        out = self.perceiver_resampler(x, clip_embeds)
        return x + scale * out if shortcut else out


class AttentionMLP(nn.Module):
    """functions.py:524-593 with the options FacialEncoder leaves at their defaults (no positional embedding, no
    mean-pooled latents)"""

    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, single_num_tokens=1, embedding_dim=1280, output_dim=768,
                 ff_mult=4):
        super().__init__()
        self.latents = nn.Parameter(torch.randn(1, single_num_tokens, dim) / dim ** 0.5)
        self.proj_in = nn.Linear(embedding_dim, dim)
        self.proj_out = nn.Linear(dim, output_dim)
        self.norm_out = nn.LayerNorm(output_dim)
        self.layers = _perceiver_layers(dim, depth, dim_head, heads, ff_mult)

    def forward(self, x):
        latents = self.latents.repeat(x.size(0), 1, 1)
        x = self.proj_in(x)
        for attn, ff in self.layers:
            latents = attn(x, latents) + latents
            latents = ff(latents) + latents
        return self.norm_out(self.proj_out(latents))


class MLP(nn.Module):
    def __init__(self, in_dim, out_dim, hidden_dim, use_residual=True):
        super().__init__()
        self.layernorm = nn.LayerNorm(in_dim)
        self.fc1 = nn.Linear(in_dim, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, out_dim)
        self.use_residual = use_residual
        self.act_fn = nn.GELU()

    def forward(self, x):
        y = self.fc2(self.act_fn(self.fc1(self.layernorm(x))))
        return y + x if self.use_residual else y


class FuseModule(nn.Module):
    def __init__(self, embed_dim):
        super().__init__()
        self.mlp1 = MLP(embed_dim * 2, embed_dim, embed_dim, use_residual=False)
        self.mlp2 = MLP(embed_dim, embed_dim, embed_dim, use_residual=True)
        self.layer_norm = nn.LayerNorm(embed_dim)

    def fuse_fn(self, prompt_embeds, id_embeds):
        s = self.mlp1(torch.cat([prompt_embeds, id_embeds], dim=-1)) + prompt_embeds
        return self.layer_norm(self.mlp2(s))

    def forward(self, prompt_embeds, id_embeds, class_tokens_mask, valid_id_mask):
        id_embeds = id_embeds.to(prompt_embeds.dtype)
        bs, seq = id_embeds.shape[0], prompt_embeds.shape[1]
        flat = id_embeds.view(-1, id_embeds.shape[-2], id_embeds.shape[-1])
        valid = flat[valid_id_mask.flatten()]
        pe = prompt_embeds.reshape(-1, prompt_embeds.shape[-1]).clone()      # (the reference scatters in place)
        mask = class_tokens_mask.view(-1)
        fused = self.fuse_fn(pe[mask], valid.view(-1, valid.shape[-1]))
        assert mask.sum() == fused.shape[0]
        pe.masked_scatter_(mask[:, None], fused.to(pe.dtype))
        return pe.view(bs, seq, -1)


class FacialEncoder(nn.Module):
    def __init__(self, embedding_dim=1280, output_dim=768, embed_dim=768):
        super().__init__()
        self.visual_projection = AttentionMLP(embedding_dim=embedding_dim, output_dim=output_dim)
        self.fuse_module = FuseModule(embed_dim=embed_dim)

    def forward(self, prompt_embeds, multi_image_embeds, class_tokens_mask, valid_id_mask):
        bs, n, t, d = multi_image_embeds.shape
        id_embeds = self.visual_projection(multi_image_embeds.view(bs * n, t, d)).view(bs, n, 1, -1)
        return self.fuse_module(prompt_embeds, id_embeds, class_tokens_mask, valid_id_mask)


@torch.no_grad()
def assemble_prompt_embeds(image_proj: ProjPlusModel, facial_encoder: FacialEncoder, *, text_embeds, negative_embeds,
                           text_only_embeds, faceid_embeds, clip_embeds, uncond_clip_embeds, facial_embeds,
                           uncond_facial_embeds, facial_token_mask, valid_facial_mask, s_scale=1.0, shortcut=False):
    """What the reference's __call__ builds between :479 and :507 from the encoders' outputs:
    ``cat([null, augmented, text_only])`` of shape [3B, 77 + num_tokens, Dc]."""
    tok = image_proj(faceid_embeds, clip_embeds, shortcut=shortcut, scale=s_scale)                          # :197
    utok = image_proj(torch.zeros_like(faceid_embeds), uncond_clip_embeds, shortcut=shortcut, scale=s_scale)  # :198
    facial = facial_encoder(text_embeds, facial_embeds, facial_token_mask, valid_facial_mask)               # :190
    ufacial = facial_encoder(negative_embeds, uncond_facial_embeds, facial_token_mask, valid_facial_mask)   # :193
    augmented = torch.cat([facial, tok], dim=1)                                                             # :492
    null = torch.cat([ufacial, utok], dim=1)                                                                # :493
    text_only = torch.cat([text_only_embeds, tok], dim=1)                                                   # :504
    return torch.cat([null, augmented, text_only], dim=0)                                                   # :495-505
