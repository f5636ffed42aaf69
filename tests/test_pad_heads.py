"""Host-side head padding (engine/towers.py::_pad_heads): checkpoints whose attention heads are not 64 / 128 wide are loaded
with zero-padded Q / K / V rows and out-projection columns.  Pure weight algebra, checked on the CPU against torch's own
multi-head attention arithmetic (what open_clip's nn.MultiheadAttention / transformers' BertSelfAttention compute)."""
import pytest
import torch

from marqo_amd.engine import towers as T


def _mha(x, qkv_w, qkv_b, out_w, heads, hd, scale):
    n, W = x.shape
    q, k, v = (x @ qkv_w.t() + qkv_b).view(n, 3, heads, hd).permute(1, 2, 0, 3)
    p = torch.softmax(q @ k.transpose(1, 2) * scale, dim=-1)
    return (p @ v).transpose(0, 1).reshape(n, heads * hd) @ out_w.t()


@pytest.mark.parametrize("d,hp", [(32, 64), (16, 64), (48, 64), (80, 96), (88, 96), (104, 112), (120, 128)])
def test_pad_heads_preserves_attention(d, hp):
    heads, n = 4, 9
    W = heads * d
    g = torch.Generator().manual_seed(d)
    x = torch.randn(n, W, generator=g, dtype=torch.float64)
    qkv_w = torch.randn(3 * W, W, generator=g, dtype=torch.float64) * W ** -0.5
    qkv_b = torch.randn(3 * W, generator=g, dtype=torch.float64) * 0.1
    out_w = torch.randn(W, W, generator=g, dtype=torch.float64) * W ** -0.5
    ref = _mha(x, qkv_w, qkv_b, out_w, heads, d, d ** -0.5)
    assert T._kernel_head_dim(d, heads) == hp and T._head_dim(W, heads) == d
    w2, b2, o2 = T._pad_heads(qkv_w, qkv_b, out_w, heads, d)
    assert w2.shape == (3 * heads * hp, W) and b2.shape == (3 * heads * hp,) and o2.shape == (W, heads * hp)
    got = _mha(x.float(), w2, b2, o2, heads, hp, hp ** -0.5)   # the kernel's scale is 1/sqrt(hp)
    assert torch.allclose(got.double(), ref, atol=1e-4, rtol=1e-4)


def test_encoder_cfg_attention_width():
    assert T._encoder_cfg(768, 12, 12, 3072, False, False, 0, 1e-5).attn_width == 0          # 64-wide heads: nothing padded
    assert T._encoder_cfg(384, 12, 12, 1536, False, True, 0, 1e-12).attn_width == 12 * 64    # e5-small
    assert T._encoder_cfg(1280, 32, 16, 5120, False, False, 0, 1e-5).attn_width == 16 * 96   # ViT-H-14: 80 -> 96
    assert T._encoder_cfg(1664, 48, 16, 8192, False, False, 0, 1e-5).attn_width == 16 * 112  # ViT-bigG-14: 104 -> 112
    assert T._encoder_cfg(240, 2, 3, 960, False, False, 0, 1e-5).attn_width == 3 * 128        # 3 heads of 80: 3 * 96 is not a multiple of 64
    assert T._encoder_cfg(2048, 2, 16, 8192, False, False, 0, 1e-5).attn_width == 0          # native 128-wide heads
    with pytest.raises(ValueError):
        T._encoder_cfg(2048, 2, 8, 8192, False, False, 0, 1e-5)                               # 256-wide heads
    with pytest.raises(ValueError):
        T._encoder_cfg(100, 2, 3, 400, False, False, 0, 1e-5)


def test_pad_mlp_is_exact():
    """MLP hidden sizes that are not a multiple of 64 (ViT-SO400M: 4304 -> 4352): zero rows / columns change nothing"""
    g = torch.Generator().manual_seed(0)
    W, F = 24, 100
    x = torch.randn(5, W, generator=g, dtype=torch.float64)
    w1, b1, w2 = torch.randn(F, W, generator=g, dtype=torch.float64), torch.randn(F, generator=g, dtype=torch.float64), torch.randn(W, F, generator=g, dtype=torch.float64)
    p1, pb, p2 = T._pad_mlp(w1, b1, w2)
    assert p1.shape == (128, W) and pb.shape == (128,) and p2.shape == (W, 128)
    for act in (torch.nn.functional.gelu, lambda t: t * torch.sigmoid(1.702 * t)):
        ref = act(x @ w1.t() + b1) @ w2.t()
        got = act(x.float() @ p1.t() + pb) @ p2.t()
        assert torch.allclose(got.double(), ref, atol=1e-4, rtol=1e-4)
    same = T._pad_mlp(torch.zeros(128, W), torch.zeros(128), torch.zeros(W, 128))
    assert same[0].shape == (128, W)
