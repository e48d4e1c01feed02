"""MI355X: A2, the fused query-side launch (csrc/mla_absorb.hip: absorption bmm + RoPE + K5 + K4), through the C-ABI, against the
four-launch chain it replaces — `bmm` (B2), `apply_rope_with_cos_sin_cache_inplace` with `output_q_rope` (R2, bit-exact vs the
reference's forward_native golden), `quantize_q_and_cache_k` (K5 + K4, bit-exact vs the reference pool golden): every output byte
must be identical (srt/models/deepseek_v2.py:830-861, flashmla_backend.py:188-206)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("T,H,neox", [(1, 16, False), (37, 16, False), (128, 128, False), (300, 16, True), (129, 32, False)])
def test_fused_absorb_rope_quant_is_bit_identical_to_the_four_launch_chain(T, H, neox):
    import os

    import flash_mla_fp8 as fm
    from fluent_mi355.bmm import bmm
    from fluent_mi355.rope import apply_rope_with_cos_sin_cache_inplace

    g = torch.Generator().manual_seed(7 * T + H)
    q = (torch.randn(T, H, 192, generator=g) * torch.exp(torch.randn(T, H, 1, generator=g))).to(torch.bfloat16).to(DEV)
    w = (torch.randn(H, 512, 128, generator=g) * 0.05).to(torch.bfloat16).to(DEV)      # k-contiguous storage
    w_kc = w.transpose(1, 2)                                                           # [H, 128, 512] as the model holds it (:1632)
    latent = (torch.randn(T, 576, generator=g) * 2).to(torch.bfloat16).to(DEV)
    max_pos = 4096
    cache = torch.randn(max_pos, 64, generator=g).to(DEV)                              # any values: the kernels only multiply by them
    pos = torch.randint(0, max_pos, (T,), generator=g).to(DEV)
    slots = 2048
    loc = torch.randperm(slots, generator=g)[:T].to(torch.int32)
    if T > 2:
        loc[1] = -1
        loc[2] = slots + 3
    loc = loc.to(DEV)

    def caches():
        return [torch.zeros(slots, 1, 512, dtype=torch.uint8, device=DEV), torch.zeros(slots, 1, 1, device=DEV),
                torch.zeros(slots, 1, 64, dtype=torch.bfloat16, device=DEV)]

    # ---- the chain ----
    lat_a = latent.clone()
    Q = torch.empty(T, H, 576, dtype=torch.bfloat16, device=DEV)
    bmm(q[..., :128].transpose(0, 1), w_kc, out=Q[..., :512].transpose(0, 1))
    K = lat_a.unsqueeze(1)
    apply_rope_with_cos_sin_cache_inplace(pos, q[..., 128:], K[..., 512:], 64, cache, is_neox=neox, output_q_rope=Q[..., 512:])
    ca = caches()
    rn, rs, rr = fm.quantize_q_and_cache_k(Q, K.contiguous(), ca[0], ca[1], ca[2], loc, 512)
    # ---- one launch ----
    lat_b = latent.clone()
    cb = caches()
    qn, qs, qr = fm.absorb_rope_quant(q, w_kc, pos, cache, lat_b, cb[0], cb[1], cb[2], loc, is_neox=neox)
    torch.cuda.synchronize()
    assert torch.equal(lat_a.view(torch.int16), lat_b.view(torch.int16))               # k_pe rotated in place, k_nope untouched
    assert torch.equal(qs, rs)
    assert torch.equal(qn.view(torch.uint8), rn.view(torch.uint8))
    assert torch.equal(qr.view(torch.int16), rr.view(torch.int16))
    for x, y in zip(ca, cb):
        assert torch.equal(x.view(torch.uint8), y.view(torch.uint8))
    # the q-only form (no K rows)
    qn2, qs2, qr2 = fm.absorb_rope_quant(q, w_kc, pos, cache, is_neox=neox)
    assert torch.equal(qn2.view(torch.uint8), rn.view(torch.uint8)) and torch.equal(qs2, rs) and torch.equal(qr2.view(torch.int16), rr.view(torch.int16))


def test_fused_absorb_refuses_what_it_does_not_serve():
    import flash_mla_fp8 as fm
    q = torch.zeros(4, 8, 192, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(8, 128, 512, dtype=torch.bfloat16, device=DEV)                     # contiguous [H, 128, 512]: not k-contiguous
    cache = torch.zeros(16, 64, device=DEV)
    pos = torch.zeros(4, dtype=torch.int64, device=DEV)
    with pytest.raises(RuntimeError):
        fm.absorb_rope_quant(q, w, pos, cache)
    with pytest.raises(RuntimeError):
        fm.absorb_rope_quant(q.float(), w.transpose(1, 2).contiguous().transpose(1, 2), pos, cache)
