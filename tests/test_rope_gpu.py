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


@pytest.mark.parametrize("T,H,loc_dtype", [(1, 128, torch.int64), (128, 16, torch.int64), (37, 8, torch.int32)])
def test_rope_absorb_prepare_call_output_q_rope_and_fused_set_kv(T, H, loc_dtype):
    """The call forward_absorb_prepare makes (models/deepseek_v2.py:843-861): q_pe = q[..., 128:] rotated INTO the rope
    columns of the absorbed Q [T, H, 576] (output_q_rope), k_pe rotated in place in K = latent[:, None, :], and — bf16 KV
    cache — FusedSetKVBufferArg as models/utils.py:52-81 builds it from the MLA pool's [slots, 1, 576] buffer:
    kv[loc, :512] = k_nope, kv[loc, 512:] = rotated k_pe.  Expected values: the in-place kernel (bit-exact vs the reference
    golden above) + MLATokenToKVPool.set_kv_buffer's `buf[loc] = cat(k_nope, k_rope)`."""
    import flashinfer

    g = torch.Generator().manual_seed(T * 7 + H)
    q = torch.randn(T, H, 192, generator=g).to(torch.bfloat16).to(DEV)
    latent = torch.randn(T, 576, generator=g).to(torch.bfloat16).to(DEV)
    cache = torch.randn(1000, 64, generator=g).to(DEV)
    pos = torch.randint(0, 1000, (T,), generator=g).to(DEV)
    slots = 4 * T + 5
    loc = torch.randperm(slots, generator=g)[:T].to(loc_dtype).to(DEV)
    # expected: in-place call on copies
    q_ref, l_ref = q.clone(), latent.clone()
    flashinfer.apply_rope_with_cos_sin_cache_inplace(pos, q_ref[..., 128:], l_ref[:, 512:].unsqueeze(1), 64, cache, is_neox=False)
    kv_ref = torch.full((slots, 1, 576), 7.0, dtype=torch.bfloat16, device=DEV)
    kv_ref[loc.long(), 0] = l_ref
    # the decode-path call
    Q = torch.full((T, H, 576), -3.0, dtype=torch.bfloat16, device=DEV)
    K = latent.clone().unsqueeze(1)
    kv = torch.full((slots, 1, 576), 7.0, dtype=torch.bfloat16, device=DEV)
    q_in = q.clone()
    arg = flashinfer.FusedSetKVBufferArg(value=K[..., :512], k_buffer=kv[..., 512:].view(slots, -1), v_buffer=kv[..., :512].view(slots, -1),
                                         k_scale=None, v_scale=None, cache_loc=loc)
    flashinfer.apply_rope_with_cos_sin_cache_inplace(pos, q_in[..., 128:], K[..., 512:], 64, cache, is_neox=False,
                                                     fused_set_kv_buffer_arg=arg, output_q_rope=Q[..., 512:])
    torch.cuda.synchronize()
    assert torch.equal(Q[..., 512:], q_ref[..., 128:]) and bool((Q[..., :512] == -3.0).all())
    assert torch.equal(q_in, q)                                   # the source of a separate output is left untouched
    assert torch.equal(K[:, 0], l_ref)                            # k_pe rotated in place, k_nope untouched
    assert torch.equal(kv, kv_ref)                                # cache rows written, every other row untouched
    with pytest.raises(RuntimeError, match="bf16 cache"):
        arg.k_scale = 0.5
        flashinfer.apply_rope_with_cos_sin_cache_inplace(pos, q_in[..., 128:], K[..., 512:], 64, cache, is_neox=False,
                                                         fused_set_kv_buffer_arg=arg)


def test_fused_set_kv_skips_negative_cache_locations():
    """a negative cache_loc entry (padded request) rotates k_pe in place but writes no cache row"""
    import flashinfer

    g = torch.Generator().manual_seed(2)
    T, H, slots = 6, 8, 20
    q = torch.randn(T, H, 192, generator=g).to(torch.bfloat16).to(DEV)
    latent = torch.randn(T, 576, generator=g).to(torch.bfloat16).to(DEV)
    cache = torch.randn(100, 64, generator=g).to(DEV)
    pos = torch.randint(0, 100, (T,), generator=g).to(DEV)
    loc = torch.tensor([3, -1, 7, 0, -1, 19], dtype=torch.int64, device=DEV)
    kv = torch.full((slots, 1, 576), 5.0, dtype=torch.bfloat16, device=DEV)
    K = latent.clone().unsqueeze(1)
    arg = flashinfer.FusedSetKVBufferArg(value=K[..., :512], k_buffer=kv[..., 512:].view(slots, -1), v_buffer=kv[..., :512].view(slots, -1),
                                         k_scale=None, v_scale=None, cache_loc=loc)
    flashinfer.apply_rope_with_cos_sin_cache_inplace(pos, q[..., 128:], K[..., 512:], 64, cache, is_neox=False, fused_set_kv_buffer_arg=arg)
    torch.cuda.synchronize()
    written = [3, 7, 0, 19]
    src = [0, 2, 3, 5]
    assert torch.equal(kv[written, 0], K[src, 0])
    untouched = [i for i in range(slots) if i not in written]
    assert bool((kv[untouched] == 5.0).all())
