"""MI355X: R2 rotary embedding (`flashinfer.apply_rope_with_cos_sin_cache_inplace`, csrc/rope.hip) through the C-ABI:
bit-exact against the golden vectors of the reference's DeepseekScalingRotaryEmbedding.forward_native and against the
oracle on the strided q_pe / k_pe views of the MLA path."""
import numpy as np
import pytest
import torch

from helpers import bf16_from_u16, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def bits(t):
    return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


@pytest.mark.parametrize("name,neox", [("gptj", False), ("neox", True)])
def test_rope_bit_exact_vs_reference_golden(name, neox):
    import flashinfer

    g = load_golden("rope_deepseek_yarn.npz")
    q, k = bf16_from_u16(g[name + "_q"]).to(DEV), bf16_from_u16(g[name + "_k"]).to(DEV)
    flashinfer.apply_rope_with_cos_sin_cache_inplace(torch.from_numpy(g[name + "_pos"]).to(DEV), q, k, 64,
                                                     torch.from_numpy(g[name + "_cache"]).to(DEV), is_neox=neox)
    torch.cuda.synchronize()
    assert np.array_equal(bits(q), g[name + "_q_out"])
    assert np.array_equal(bits(k), g[name + "_k_out"])


@pytest.mark.parametrize("T,H,neox", [(1, 128, False), (128, 128, False), (4097, 16, False), (33, 8, True)])
def test_rope_on_strided_mla_views_vs_oracle(T, H, neox):
    """q_pe = q[..., 128:] of q [T, H, 192] and k_pe = latent[:, None, 512:] of the latent row [T, 576]
    (deepseek_v2.py:636-647): rotated in place, every other element untouched (bit-exact)."""
    import flashinfer
    from oracle import rope_ref

    g = torch.Generator().manual_seed(T + H)
    q = torch.randn(T, H, 192, generator=g).to(torch.bfloat16)
    latent = torch.randn(T, 576, generator=g).to(torch.bfloat16)
    cache = torch.randn(1000, 64, generator=g)
    pos = torch.randint(0, 1000, (T,), generator=g)
    qd, ld = q.to(DEV), latent.to(DEV)
    flashinfer.apply_rope_with_cos_sin_cache_inplace(pos.to(DEV), qd[..., 128:], ld[:, 512:].unsqueeze(1), 64, cache.to(DEV),
                                                     is_neox=neox)
    torch.cuda.synchronize()
    q_ref = bits(q).copy()
    q_ref[..., 128:] = rope_ref.apply_rope(pos.numpy(), bits(q)[..., 128:], cache.numpy(), neox)
    l_ref = bits(latent).copy()
    l_ref[:, 512:] = rope_ref.apply_rope(pos.numpy(), bits(latent)[:, None, 512:], cache.numpy(), neox)[:, 0]
    assert np.array_equal(bits(qd), q_ref)
    assert np.array_equal(bits(ld), l_ref)


def test_rope_flat_layout_partial_rotary_and_bad_arguments():
    import flashinfer
    from oracle import rope_ref

    g = torch.Generator().manual_seed(5)
    T, H, D, R = 9, 4, 128, 64   # rotary_dim < head_size: the tail passes through (rotary_embedding.py:817-820)
    q = torch.randn(T, H * D, generator=g).to(torch.bfloat16)
    k = torch.randn(T, 2 * D, generator=g).to(torch.bfloat16)
    cache = torch.randn(50, R, generator=g)
    pos = torch.randint(0, 50, (T,), generator=g)
    qd, kd = q.to(DEV), k.to(DEV)
    flashinfer.apply_rope_with_cos_sin_cache_inplace(pos.to(DEV), qd, kd, D, cache.to(DEV), is_neox=True)
    torch.cuda.synchronize()
    assert np.array_equal(bits(qd).reshape(T, H, D), rope_ref.apply_rope(pos.numpy(), bits(q).reshape(T, H, D), cache.numpy(), True))
    assert np.array_equal(bits(kd).reshape(T, 2, D), rope_ref.apply_rope(pos.numpy(), bits(k).reshape(T, 2, D), cache.numpy(), True))
    flashinfer.apply_rope_with_cos_sin_cache_inplace(pos[:0].to(DEV), qd[:0], kd[:0], D, cache.to(DEV))   # empty batch
    with pytest.raises(RuntimeError):
        flashinfer.apply_rope_with_cos_sin_cache_inplace(pos.to(DEV), qd, kd, D, cache.to(DEV).to(torch.bfloat16))
    with pytest.raises(RuntimeError):
        flashinfer.apply_rope_with_cos_sin_cache_inplace(pos.to(DEV), qd.float(), kd, D, cache.to(DEV))
    with pytest.raises(RuntimeError):
        flashinfer.apply_rope_with_cos_sin_cache_inplace(pos[:3].to(DEV), qd, kd, D, cache.to(DEV))
