"""Attention + processors, restating diffusers 0.24.0."""
from typing import Union

import torch
import torch.nn.functional as F
from torch import nn


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, out_bias=True, processor=None, **unused):
        super().__init__()
        self.inner_dim = dim_head * heads
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.upcast_attention = upcast_attention
        self.upcast_softmax = upcast_softmax
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(dropout)])
        self.processor = processor if processor is not None else AttnProcessor2_0()

    def set_processor(self, processor):
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask)

    def head_to_batch_dim(self, t):
        b, l, d = t.shape
        return t.reshape(b, l, self.heads, d // self.heads).permute(0, 2, 1, 3).reshape(b * self.heads, l, d // self.heads)

    def batch_to_head_dim(self, t):
        bh, l, d = t.shape
        b = bh // self.heads
        return t.reshape(b, self.heads, l, d).permute(0, 2, 1, 3).reshape(b, l, d * self.heads)


class AttnProcessor:
    """softmax(QK^T * scale) V through explicit bmm (the pre-SDPA processor)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        assert attention_mask is None
        ehs = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q = attn.head_to_batch_dim(attn.to_q(hidden_states))
        k = attn.head_to_batch_dim(attn.to_k(ehs))
        v = attn.head_to_batch_dim(attn.to_v(ehs))
        probs = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * attn.scale, dim=-1)
        out = attn.batch_to_head_dim(torch.bmm(probs, v))
        out = attn.to_out[0](out)
        return attn.to_out[1](out)


class AttnProcessor2_0:
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        assert attention_mask is None
        ehs = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        b = hidden_states.shape[0]
        q, k, v = attn.to_q(hidden_states), attn.to_k(ehs), attn.to_v(ehs)
        hd = q.shape[-1] // attn.heads
        q = q.view(b, -1, attn.heads, hd).transpose(1, 2)
        k = k.view(b, -1, attn.heads, hd).transpose(1, 2)
        v = v.view(b, -1, attn.heads, hd).transpose(1, 2)
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)
        out = out.transpose(1, 2).reshape(b, -1, attn.heads * hd).to(q.dtype)
        out = attn.to_out[0](out)
        return attn.to_out[1](out)


AttentionProcessor = Union[AttnProcessor, AttnProcessor2_0]


class AttnAddedKVProcessor:
    pass


ADDED_KV_ATTENTION_PROCESSORS = (AttnAddedKVProcessor,)
CROSS_ATTENTION_PROCESSORS = (AttnProcessor, AttnProcessor2_0)
