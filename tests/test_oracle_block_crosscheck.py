"""Cross-check the restated timm ViT block against an INDEPENDENT implementation.

timm is not under /root/reference and is not installed, so the block is "parity
unpinned" by the reference; transformers' Dinov2Layer implements the same
LN -> attn -> LayerScale -> residual / LN -> MLP -> LayerScale -> residual block.
"""

import pytest
import torch

from oracle.encoder import Block


def test_block_matches_dinov2_layer():
    tr = pytest.importorskip("transformers")
    from transformers.models.dinov2.modeling_dinov2 import Dinov2Layer

    dim, heads = 128, 2
    cfg = tr.Dinov2Config(hidden_size=dim, num_attention_heads=heads, mlp_ratio=4,
                          layer_norm_eps=1e-5, layerscale_value=1.0, qkv_bias=True,
                          drop_path_rate=0.0, hidden_act="gelu", use_swiglu_ffn=False,
                          attn_implementation="eager")
    ref = Dinov2Layer(cfg).eval()
    blk = Block(dim, heads, 4.0, 0.0, 1e-5).eval()
    g = torch.Generator().manual_seed(0)
    for p in blk.parameters():
        p.data = torch.randn(p.shape, generator=g) * 0.2
    sd = blk.state_dict()
    q, k, v = sd["attn.qkv.weight"].chunk(3, 0)
    qb, kb, vb = sd["attn.qkv.bias"].chunk(3, 0)
    att = ref.attention.attention
    att.query.weight.data, att.key.weight.data, att.value.weight.data = q, k, v
    att.query.bias.data, att.key.bias.data, att.value.bias.data = qb, kb, vb
    ref.attention.output.dense.weight.data = sd["attn.proj.weight"]
    ref.attention.output.dense.bias.data = sd["attn.proj.bias"]
    ref.norm1.load_state_dict({"weight": sd["norm1.weight"], "bias": sd["norm1.bias"]})
    ref.norm2.load_state_dict({"weight": sd["norm2.weight"], "bias": sd["norm2.bias"]})
    ref.mlp.fc1.load_state_dict({"weight": sd["mlp.fc1.weight"], "bias": sd["mlp.fc1.bias"]})
    ref.mlp.fc2.load_state_dict({"weight": sd["mlp.fc2.weight"], "bias": sd["mlp.fc2.bias"]})
    ref.layer_scale1.lambda1.data = sd["ls1.gamma"]
    ref.layer_scale2.lambda1.data = sd["ls2.gamma"]
    x = torch.randn(2, 17, dim, generator=g)
    with torch.no_grad():
        a = blk(x)
        b = ref(x)
        b = b[0] if isinstance(b, tuple) else b
    assert torch.allclose(a, b, atol=2e-5, rtol=1e-5), (a - b).abs().max()
