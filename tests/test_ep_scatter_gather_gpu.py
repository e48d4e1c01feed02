"""MI355X: C8/C9 ep_scatter / ep_gather of the DeepExecutor (`fluent_mi355.ep`, csrc/ep_scatter_gather.hip) through the
C-ABI against golden vectors produced by the reference's own Triton kernels (run by Triton's CPU interpreter) and against
the oracle at decode / prefill sizes.  The reference hands out rows inside an expert's group with atomics: the order inside
a group is unspecified, so groups are compared as sets and output_index by what it points at."""
import numpy as np
import pytest
import torch

from helpers import bf16_from_u16, load_golden
from oracle import ep_ref
from oracle.rope_ref import bf16_to_f32

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def run_scatter(x, xs, topk, padded):
    from fluent_mi355.ep import ep_scatter

    T, K = topk.shape
    E, M, H = padded.shape[0], int(padded.sum()), x.shape[1]
    start = torch.zeros(E, dtype=torch.int32, device=DEV)
    out = torch.zeros(M, H, dtype=torch.uint8, device=DEV).view(torch.float8_e4m3fn)
    outs = torch.zeros(M, H // 128, device=DEV)
    m_idx = torch.full((M,), -1, dtype=torch.int32, device=DEV)
    oidx = torch.full((T, K), -1, dtype=torch.int32, device=DEV)
    ep_scatter(torch.from_numpy(x).to(DEV).view(torch.float8_e4m3fn), torch.from_numpy(xs).to(DEV), torch.from_numpy(topk).to(DEV),
               torch.from_numpy(padded).to(DEV), start, out, outs, m_idx, oidx)
    torch.cuda.synchronize()
    return start.cpu().numpy(), out.view(torch.uint8).cpu().numpy(), outs.cpu().numpy(), m_idx.cpu().numpy(), oidx.cpu().numpy()


def check_scatter(got, want, x, xs, topk, padded):
    start, out, outs, m_idx, oidx = got
    w_start, w_out, w_outs, w_m_idx, w_oidx = want
    assert np.array_equal(start, w_start) and np.array_equal(m_idx, w_m_idx)
    assert np.array_equal(oidx >= 0, w_oidx >= 0)
    begin = np.cumsum(padded) - padded
    T, K = topk.shape
    for t in range(T):
        for k in range(K):
            e = int(topk[t, k])
            if e >= 0:
                d = int(oidx[t, k])
                assert begin[e] <= d < w_start[e]                          # inside the filled part of its expert's group
                assert np.array_equal(out[d], x[t]) and np.array_equal(outs[d], xs[t])
    live = oidx[oidx >= 0]
    assert np.unique(live).size == live.size                               # every pair got its own row
    untouched = np.ones(out.shape[0], bool); untouched[live] = False
    assert not out[untouched].any() and not outs[untouched].any()          # padding rows are not written
    for e in range(padded.shape[0]):                                       # same multiset of rows per group as the reference
        a, b = out[begin[e]:w_start[e]], w_out[begin[e]:w_start[e]]
        assert np.array_equal(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])])


def test_ep_scatter_and_gather_vs_reference_triton_golden():
    from fluent_mi355.ep import ep_gather

    g = load_golden("ep_scatter_gather.npz")
    got = run_scatter(g["x"], g["xs"], g["topk"], g["padded"])
    check_scatter(got, (g["start_after"], g["out"], g["outs"], g["m_idx"], g["oidx"]), g["x"], g["xs"], g["topk"], g["padded"])
    out = torch.zeros(g["topk"].shape[0], g["y"].shape[1], dtype=torch.bfloat16, device=DEV)
    ep_gather(bf16_from_u16(g["y"]).to(DEV), torch.from_numpy(g["topk"]).to(DEV), torch.from_numpy(g["w"]).to(DEV),
              torch.from_numpy(g["oidx"]).to(DEV), out)
    want = torch.from_numpy(g["gathered"]).to(torch.bfloat16)              # the reference's fp32 sums, rounded once
    diff = (out.cpu().float() - want.float()).abs()
    assert float((diff / (want.float().abs() + 1e-3)).max()) < 1e-2         # <= 1 bf16 ulp (fma contraction on the device)
    assert float((out.cpu().view(torch.int16) != want.view(torch.int16)).float().mean()) < 0.02


@pytest.mark.parametrize("T,H,K,E,int64_ids", [(1, 7168, 8, 32, True), (256, 7168, 8, 32, False), (3000, 2048, 6, 9, True)])
def test_ep_scatter_gather_vs_oracle(T, H, K, E, int64_ids):
    from fluent_mi355.ep import ep_gather

    rng = np.random.default_rng(T + H)
    x = rng.integers(0, 255, (T, H), dtype=np.uint8)
    xs = rng.random((T, H // 128), dtype=np.float32)
    topk = np.stack([rng.permutation(2 * E)[:K] for _ in range(T)]).astype(np.int64 if int64_ids else np.int32)
    topk[topk >= E] = -1
    cnt = np.bincount(topk[topk >= 0], minlength=E)
    padded = ((cnt + 127) // 128 * 128).astype(np.int32)
    want = ep_ref.ep_scatter(x, xs, topk, padded)
    got = run_scatter(x, xs, topk, padded)
    check_scatter(got, want, x, xs, topk, padded)
    # gather back through the DEVICE's own output_index
    M = int(padded.sum())
    y = torch.from_numpy(rng.standard_normal((M, H), dtype=np.float32)).to(torch.bfloat16)
    w = rng.random((T, K), dtype=np.float32)
    out = torch.zeros(T, H, dtype=torch.bfloat16, device=DEV)
    ep_gather(y.to(DEV), torch.from_numpy(topk).to(DEV), torch.from_numpy(w).to(DEV), torch.from_numpy(got[4]).to(DEV), out)
    ref = ep_ref.ep_gather(y.float().numpy(), topk, w, got[4])
    assert np.allclose(out.cpu().float().numpy(), ref, rtol=1e-2, atol=2e-2)


def test_ep_scatter_gather_rejects_bad_arguments():
    from fluent_mi355.ep import ep_gather, ep_scatter

    z = lambda *s, dt=torch.int32: torch.zeros(*s, dtype=dt, device=DEV)
    with pytest.raises(RuntimeError):   # m_indices not a multiple of 128 rows
        ep_scatter(z(2, 256, dt=torch.uint8), z(2, 2, dt=torch.float32), z(2, 2), z(4), z(4), z(100, 256, dt=torch.uint8),
                   z(100, 2, dt=torch.float32), z(100), z(2, 2))
    with pytest.raises(RuntimeError):   # fp32 rows are not fp8 rows
        ep_scatter(z(2, 256, dt=torch.float32), z(2, 2, dt=torch.float32), z(2, 2), z(4), z(4), z(128, 256, dt=torch.uint8),
                   z(128, 2, dt=torch.float32), z(128), z(2, 2))
    with pytest.raises(RuntimeError):
        ep_gather(z(128, 256, dt=torch.float32), z(2, 2), z(2, 2, dt=torch.float32), z(2, 2), z(2, 256, dt=torch.bfloat16))
