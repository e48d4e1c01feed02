"""MI355X: the fused kernels behind flashinfer.comm.trtllm_{allreduce,reducescatter,allgather}_fusion (csrc/norm_fused.hip)
through the drop-in API at world size 1 (the exchange is the identity; the N>1 host logic runs in test_comm_gloo_cpu.py).
bf16 outputs must match the reference's RMSNorm.forward_native golden BIT FOR BIT up to the last-ulp differences of the
reduction order (fp32 sum of squares): tolerance = 1 bf16 ulp on < 0.5 % of the elements; the FP8 bytes are compared against
the oracle's quantisation of the kernel's own bf16 norm (bit-exact)."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "sglang-fluentllm_amd"))
from helpers import bf16_from_u16, load_golden  # noqa: E402
from oracle import gemm_ref, norm_ref  # noqa: E402

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def ulp_close(a, b, frac=5e-3):
    """bf16 tensors equal except for <= 1 ulp on at most `frac` of the elements."""
    ai, bi = a.cpu().view(torch.int16).int(), b.cpu().view(torch.int16).int()
    d = (ai - bi).abs()
    return int(d.max()) <= 1 and float((d > 0).float().mean()) <= frac


@pytest.fixture(scope="module")
def comm():
    import flashinfer.comm as c
    return c


def test_allreduce_fusion_matches_reference_rmsnorm_golden(comm):
    g = load_golden("rmsnorm_native.npz")
    _, ws = comm.trtllm_create_ipc_workspace_for_all_reduce_fusion(0, 1, 64, 7168)
    for name in ("h7168", "q1536", "kv512"):
        w, x, r = (bf16_from_u16(g[f"{name}_{k}"]).to(dev()) for k in ("w", "x", "r"))
        T, H = x.shape
        norm_out, res_out = torch.empty_like(x), torch.empty_like(x)
        quant_out = torch.empty(T, H, dtype=torch.float8_e4m3fn, device=dev())
        # the executor's column-major scale view (fp8_kernel.py create_per_token_group_quant_fp8_output_scale)
        Tp = (T + 3) // 4 * 4
        scale_out = torch.empty(H // 128, Tp, dtype=torch.float32, device=dev()).permute(-1, -2)[:T]
        comm.trtllm_allreduce_fusion(allreduce_in=x, world_size=1, world_rank=0, token_num=T, hidden_dim=H, workspace_ptrs=ws,
                                     pattern_code=comm.AllReduceFusionPattern.kARResidualRMSNormFP8BlockWiseQuant,
                                     residual_in=r, residual_out=res_out, norm_out=norm_out, quant_out=quant_out,
                                     scale_out=scale_out, rms_gamma=w, rms_eps=1e-6)
        torch.cuda.synchronize()
        assert torch.equal(res_out.cpu(), bf16_from_u16(g[f"{name}_r_out"])), name          # x + residual: exact
        assert ulp_close(norm_out, bf16_from_u16(g[f"{name}_y_res"])), name
        q_ref, s_ref = gemm_ref.per_token_group_quant_fp8(norm_out.cpu().contiguous(), 128)
        assert torch.equal(quant_out.cpu().view(torch.uint8), q_ref.view(torch.uint8)), name
        assert torch.equal(scale_out.cpu().contiguous(), s_ref), name


@pytest.mark.parametrize("T,H,W", [(1, 7168, 1), (7, 7168, 8), (64, 7168, 8), (5, 2048, 2), (3, 8192, 4)])
def test_fused_add_rmsnorm_pieces(T, H, W):
    """The C-ABI kernel itself with W received pieces + add_in + residual (what C5/C6 run after the exchange)."""
    from fluent_mi355.comm import HipNormOps
    g = torch.Generator().manual_seed(T * 131 + W)
    pieces = torch.randn(W, T, H, generator=g).to(torch.bfloat16)
    add, res = (torch.randn(T, H, generator=g).to(torch.bfloat16) for _ in range(2))
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(torch.bfloat16)
    y_ref, r_ref = norm_ref.fused_add_rmsnorm(pieces, add, res, gamma, 1e-6)
    norm_out = torch.empty(T, H, dtype=torch.bfloat16, device=dev())
    res_out = torch.empty_like(norm_out)
    HipNormOps().add_rmsnorm(pieces.to(dev()), add.to(dev()), res.to(dev()), gamma.to(dev()), 1e-6, res_out, norm_out, None, None)
    torch.cuda.synchronize()
    assert ulp_close(res_out, r_ref, frac=2e-2)      # fp32 sum of W+2 terms: order-dependent last bit
    assert ulp_close(norm_out, y_ref, frac=2e-2)


@pytest.mark.gpu
def test_flashinfer_norm_entry_points_gemma_offset_and_strided_in_place():
    """flashinfer.norm: the Gemma forms add the 1 in fp32 (bf16(w + 1) sits on the 2^-7 grid: small gammas would be lost); a strided
    `out` / in-place pair is written through the CALLER's tensors, never through a reshaped copy of them."""
    import flashinfer.norm as fn
    g = torch.Generator().manual_seed(11)
    T, H = 6, 2048
    x = torch.randn(T, H, generator=g).to(torch.bfloat16)
    w = (0.003 * torch.randn(H, generator=g)).to(torch.bfloat16)          # |w| << 2^-7: invisible after bf16(w + 1)
    xf = x.float()
    ref = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * (1.0 + w.float())).to(torch.bfloat16)
    y = fn.gemma_rmsnorm(x.to(dev()), w.to(dev()), 1e-6)
    assert ulp_close(y, ref, frac=2e-2)
    assert not torch.equal(ref, (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * (w + 1.0).float()).to(torch.bfloat16))   # the test can tell
    # in place, both operands strided views of wider tensors ([T, 2, H] -> [:, 0]): 3-D and not viewable as [rows, H]
    big_x = torch.randn(T, 2, 3, H, generator=g).to(torch.bfloat16).to(dev())
    big_r = torch.randn(T, 2, 3, H, generator=g).to(torch.bfloat16).to(dev())
    xin, rin = big_x[:, 0], big_r[:, 1]
    x0, r0 = xin.clone(), rin.clone()
    keep_x, keep_r = big_x[:, 1].clone(), big_r[:, 0].clone()
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(torch.bfloat16)
    y_ref, r_ref = norm_ref.fused_add_rmsnorm(x0.reshape(1, -1, H).cpu(), None, r0.reshape(-1, H).cpu(), gamma, 1e-6)
    fn.fused_add_rmsnorm(xin, rin, gamma.to(dev()), 1e-6)
    torch.cuda.synchronize()
    assert ulp_close(rin.reshape(-1, H), r_ref, frac=2e-2) and ulp_close(xin.reshape(-1, H), y_ref, frac=2e-2)
    assert torch.equal(big_x[:, 1], keep_x) and torch.equal(big_r[:, 0], keep_r)      # the neighbours are untouched
    out = torch.zeros(T, 2, H, dtype=torch.bfloat16, device=dev())
    fn.rmsnorm(x.to(dev()), gamma.to(dev()), 1e-6, out=out[:, 1])
    torch.cuda.synchronize()
    assert ulp_close(out[:, 1], norm_ref.rmsnorm_native(x, gamma, 1e-6), frac=2e-2) and int(out[:, 0].abs().sum()) == 0


def test_reducescatter_and_allgather_fusion_world1(comm):
    g = torch.Generator().manual_seed(5)
    _, ws = comm.trtllm_create_ipc_workspace_for_all_reduce_fusion(0, 1, 64, 7168)
    T, H = 9, 7168
    x, add, res = (torch.randn(T, H, generator=g).to(torch.bfloat16) for _ in range(3))
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(torch.bfloat16)
    y_ref, r_ref = norm_ref.fused_add_rmsnorm(x.unsqueeze(0), add, res, gamma, 1e-6)
    norm_out = torch.empty(T, H, dtype=torch.bfloat16, device=dev())
    res_out = torch.empty_like(norm_out)
    comm.trtllm_reducescatter_fusion(reducescatter_in=x.to(dev()), world_size=1, world_rank=0, token_num=T, hidden_dim=H,
                                     workspace_ptrs=ws, num_token_current_rank=T,
                                     pattern_code=comm.ReduceScatterFusionPattern.kRSAddResidualRMSNorm, add_in=add.to(dev()),
                                     residual_in=res.to(dev()), residual_out=res_out, norm_out=norm_out,
                                     rms_gamma=gamma.to(dev()), rms_eps=1e-6)
    assert ulp_close(norm_out, y_ref, frac=2e-2) and ulp_close(res_out, r_ref, frac=2e-2)
    # C7: DeepSeek-V3 q_lora 1536 + kv_lora 512 + rope 64
    q_rank, kv_rank, rope = 1536, 512, 64
    D = q_rank + kv_rank + rope
    full = torch.randn(T, D, generator=g).to(torch.bfloat16)
    gq = (1 + 0.1 * torch.randn(q_rank, generator=g)).to(torch.bfloat16)
    gkv = (1 + 0.1 * torch.randn(kv_rank, generator=g)).to(torch.bfloat16)
    x_ref, ag_ref = norm_ref.dual_rmsnorm(full, q_rank, kv_rank, gq, gkv, 1e-6, 1e-6)
    ag = torch.empty(T, D, dtype=torch.bfloat16, device=dev())
    quant_out = torch.empty(T, q_rank, dtype=torch.float8_e4m3fn, device=dev())
    scale_out = torch.empty(q_rank // 128, (T + 3) // 4 * 4, dtype=torch.float32, device=dev()).permute(-1, -2)[:T]
    xn = torch.empty(T, q_rank, dtype=torch.bfloat16, device=dev())
    comm.trtllm_allgather_fusion(allgather_in=full.to(dev()), world_size=1, world_rank=0, hidden_dim=D, workspace_ptrs=ws,
                                 num_token_current_rank=T, allgather_out=ag, num_token_all_group=T,
                                 pattern_code=comm.AllGatherFusionPattern.kAllGatherfusedRMSFP8BlockWiseQuant, x_norm_out=xn,
                                 y_norm_out=ag[..., q_rank:q_rank + kv_rank], quant_out=quant_out, scale_out=scale_out,
                                 x_rms_gamma=gq.to(dev()), y_rms_gamma=gkv.to(dev()), x_rms_eps=1e-6, y_rms_eps=1e-6,
                                 q_lora_rank=q_rank, kv_lora_rank=kv_rank, qk_rope_head_dim=rope)
    torch.cuda.synchronize()
    assert ulp_close(xn, x_ref, frac=2e-2) and ulp_close(ag, ag_ref, frac=2e-2)
    assert torch.equal(ag.cpu()[:, q_rank + kv_rank:], full[:, q_rank + kv_rank:])           # rope columns untouched
    q_ref, s_ref = gemm_ref.per_token_group_quant_fp8(xn.cpu().contiguous(), 128)
    assert torch.equal(quant_out.cpu().view(torch.uint8), q_ref.view(torch.uint8)) and torch.equal(scale_out.cpu().contiguous(), s_ref)


def test_tpdp_convertor_world1_and_sum_kernel():
    """C3: eps.communication.TPDPConvertor at world 1 (identity exchange) + the sum-only mode of the fused kernel that the
    N>1 reduce-scatter runs after its exchange."""
    from eps.communication import TPDPConvertor
    from fluent_mi355.comm import HipNormOps
    H, T = 7168, 13
    conv = TPDPConvertor(TPDPConvertor.Params(0, 64, 1, H, None))
    g = torch.Generator().manual_seed(3)
    x = torch.randn(T, H, generator=g).to(torch.bfloat16).to(dev())
    rs = conv.get_reduce_scatter_context(T, 45)
    rs.input().view(dtype=torch.bfloat16).copy_(x)
    conv.reduce_scatter(rs, torch.cuda.current_stream().cuda_stream)
    assert rs.output_row_offset == 0 and torch.equal(rs.output(), x)
    ag = conv.get_all_gather_context(T, H, 28)
    ag.input().copy_(x)
    conv.all_gather(ag, torch.cuda.current_stream().cuda_stream)
    assert torch.equal(ag.output(), x)
    pieces = torch.randn(8, T, H, generator=g).to(torch.bfloat16)
    out = torch.empty(T, H, dtype=torch.bfloat16, device=dev())
    HipNormOps().add_rmsnorm(pieces.to(dev()), None, None, None, 0.0, out, None, None, None)
    torch.cuda.synchronize()
    assert ulp_close(out, pieces.float().sum(0).to(torch.bfloat16), frac=2e-2)


DEV = torch.device("cuda:0")


def _oneshot(world_hidden=7168, max_tokens=128):
    from fluent_mi355.oneshot import OneShotComm
    return OneShotComm(0, 1, max_tokens, world_hidden)


@pytest.mark.parametrize("T,H", [(1, 7168), (37, 7168), (128, 2048), (5, 136)])
def test_oneshot_world1_is_bit_identical_to_the_fused_kernel(T, H):
    """C5/C6 one-shot route (csrc/comm_oneshot.hip: push -> flags -> wait -> fused epilogue in ONE kernel) at world 1 against
    the RCCL route's kernel on the same rows: same epilogue code, same summation order -> identical bits; plus the golden
    vectors of RMSNorm.forward_native through the public entry point below."""
    from fluent_mi355.comm import HipNormOps
    ops, c = HipNormOps(), _oneshot()
    g = torch.Generator().manual_seed(T * 31 + H)
    x = torch.randn(T, H, generator=g).to(torch.bfloat16).to(DEV)
    add = torch.randn(T, H, generator=g).to(torch.bfloat16).to(DEV)
    res = torch.randn(T, H, generator=g).to(torch.bfloat16).to(DEV)
    gamma = torch.rand(H, generator=g).to(torch.bfloat16).to(DEV)
    quant = H % 128 == 0

    def outs():
        return (torch.empty(T, H, dtype=torch.bfloat16, device=DEV), torch.empty(T, H, dtype=torch.bfloat16, device=DEV),
                torch.empty(T, H, dtype=torch.float8_e4m3fn, device=DEV) if quant else None,
                torch.empty(T, H // 128, dtype=torch.float32, device=DEV) if quant else None)
    e = outs()
    ops.add_rmsnorm(x.unsqueeze(0), None, res, gamma, 1e-6, *e)
    o = outs()
    c.allreduce_fused(x, res, gamma, 1e-6, *o)
    e2 = outs()
    ops.add_rmsnorm(x.unsqueeze(0), add, res, gamma, 1e-6, *e2)
    o2 = outs()
    c.reducescatter_fused(x, add, res, gamma, 1e-6, *o2)
    s_e, s_o = torch.empty(T, H, dtype=torch.bfloat16, device=DEV), torch.empty(T, H, dtype=torch.bfloat16, device=DEV)
    ops.add_rmsnorm(x.unsqueeze(0), None, None, None, 0.0, s_e, None, None, None)
    c.allreduce_fused(x, residual_out=s_o)
    c.check()
    for a, b in list(zip(e, o)) + list(zip(e2, o2)) + [(s_e, s_o)]:
        if a is not None:
            assert torch.equal(a.view(torch.uint8) if a.dtype == torch.float8_e4m3fn else a, b.view(torch.uint8) if b.dtype == torch.float8_e4m3fn else b)
    c.close()


def test_oneshot_replays_in_a_hipgraph_and_rejects_oversize():
    """the epoch lives in device memory: a captured launch replays (different data each replay); sizes beyond the workspace
    are refused loudly"""
    from fluent_mi355.comm import HipNormOps
    ops, c = HipNormOps(), _oneshot(2048, 32)
    T, H = 32, 2048
    x = torch.zeros(T, H, dtype=torch.bfloat16, device=DEV)
    res = torch.zeros(T, H, dtype=torch.bfloat16, device=DEV)
    gamma = torch.ones(H, dtype=torch.bfloat16, device=DEV)
    o_res, o_norm = torch.empty_like(x), torch.empty_like(x)
    c.allreduce_fused(x, res, gamma, 1e-6, o_res, o_norm)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        c.allreduce_fused(x, res, gamma, 1e-6, o_res, o_norm)
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        c.allreduce_fused(x, res, gamma, 1e-6, o_res, o_norm)
        c.reducescatter_fused(o_norm, None, None, gamma, 1e-6, None, o_res)      # two dependent operations per replay
    for it in range(4):
        g = torch.Generator().manual_seed(it)
        x.copy_(torch.randn(T, H, generator=g).to(torch.bfloat16))
        res.copy_(torch.randn(T, H, generator=g).to(torch.bfloat16))
        gr.replay()
        e_res, e_norm, e2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        ops.add_rmsnorm(x.unsqueeze(0), None, res, gamma, 1e-6, e_res, e_norm, None, None)
        ops.add_rmsnorm(e_norm.unsqueeze(0), None, None, gamma, 1e-6, None, e2, None, None)
        assert torch.equal(o_norm, e_norm) and torch.equal(o_res, e2), it
    c.check()
    with pytest.raises(RuntimeError, match="exceeds"):
        c.allreduce_fused(torch.zeros(33, H, dtype=torch.bfloat16, device=DEV), res, gamma, 1e-6, o_res, o_norm)
    c.close()


def test_oneshot_two_processes_on_one_gpu_hipipc():
    """The multi-rank kernel path on hardware, as far as a one-GPU box allows: two PROCESSES, each with its own workspace,
    map each other's through hipIpc and run fused all-reduces / reduce-scatters against each other (real cross-process
    flag waits; handles travel over gloo).  Bit-identical to the RCCL route's kernel on the stacked inputs.  Every wait is
    time-bounded: a co-scheduling problem would be an error, not a hang."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "oneshot_two_procs.py"), "2"], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("parity OK") == 2, r.stdout[-2000:]


@pytest.mark.parametrize("world", [2, 3])
def test_ep_exchange_on_the_oneshot_transport_between_processes_on_one_gpu(world):
    """C1 / C2 (eps.fast_ep.AllToAll.dispatch / combine) with every exchange ONE launch of the peer-mapped transport
    (oneshot_a2a_kernel: rows pushed into the peers' hipIpc-mapped inboxes, per-row flags, empty slab rows as their 64-byte tail):
    2 and 3 processes on the one GPU, bit-identical to the same host logic + row kernels with the exchange staged through a gloo
    all_to_all_single — random routing, a third of the capacity, every token to one peer (full slab), with and without the routing
    weights in the dispatch message; no RCCL call is counted."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "oneshot_two_procs.py"), str(world), "ep"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("one-shot transport OK") == world, r.stdout[-2000:]


def test_ep_exchange_world1_oneshot_route_matches_the_plain_route(monkeypatch):
    """FLUENT_ONESHOT=1 builds the communicator at world 1: dispatch / combine then run through oneshot_a2a_kernel (a self-push through
    the inbox) and must give the bytes of the world-1 plain route (the slab a rank sends is the slab it receives), also when replayed
    from a hipGraph (the epoch lives in device memory)."""
    from eps.fast_ep import AllToAll
    E, K, HID, t = 32, 8, 7168, 29
    g = torch.Generator().manual_seed(3)
    x = torch.randn(t, HID, generator=g).to(torch.bfloat16).to(DEV)
    idx = torch.stack([torch.randperm(E, generator=g)[:K] for _ in range(t)]).to(torch.int32).to(DEV)
    w = torch.rand(t, K, generator=g).to(DEV)

    def run(a2a):
        ex = torch.zeros(E + 1, dtype=torch.int32, device=DEV)
        xr = torch.zeros(a2a.cap * K, HID, dtype=torch.bfloat16, device=DEV)
        a2a.dispatch(ex, xr, x, idx, t, weights=w)
        y = (xr.float() * 0.5 - 1.0).to(torch.bfloat16)
        out = torch.zeros(t, HID, dtype=torch.bfloat16, device=DEV)
        a2a.combine(out, w, y, t)
        return out, ex

    plain = AllToAll(K, E, HID, 32, None)
    assert plain.oneshot is None
    monkeypatch.setenv("FLUENT_ONESHOT", "1")
    assert AllToAll(K, E, HID, 32, None).oneshot is None      # the EP route on the one-shot transport is opt-in (FLUENT_EP_ONESHOT=1)
    monkeypatch.setenv("FLUENT_EP_ONESHOT", "1")
    one = AllToAll(K, E, HID, 32, None)
    assert one.oneshot is not None and "one-shot" in one.comm_route
    o0, e0 = run(plain)
    o1, e1 = run(one)
    torch.cuda.synchronize()
    assert torch.equal(o0, o1) and torch.equal(e0, e1)
    assert one.messages == {"oneshot": 2, "rccl": 0}
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run(one)
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        o2, _ = run(one)
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    one.oneshot.check()
    assert torch.equal(o2, o0)


def test_oneshot_allgather_and_dual_rmsnorm_world1_match_the_collective_route(monkeypatch):
    """C7 / C3 on the one-shot transport at world 1 (FLUENT_ONESHOT=1): flashinfer.comm.trtllm_allgather_fusion (gather only, and
    gather + dual RMSNorm + fp8 quant) and TPDPConvertor.reduce_scatter / all_gather give the bits of the RCCL-route kernels;
    vllm_ar.all_reduce sums in place through fl_allreduce_fused."""
    import flashinfer.comm as comm
    from flashinfer.comm import vllm_ar
    from fluent_mi355.comm import HipNormOps, TPDPConvertor
    T, D, QR, KVR = 37, 2112, 1536, 512
    g = torch.Generator().manual_seed(21)
    x = torch.randn(T, D, generator=g).to(torch.bfloat16).to(DEV)
    gq, gkv = torch.rand(QR, generator=g).to(torch.bfloat16).to(DEV), torch.rand(KVR, generator=g).to(torch.bfloat16).to(DEV)
    got = []
    for flag in ("0", "1"):
        monkeypatch.setenv("FLUENT_ONESHOT", flag)
        handles, wsp = comm.trtllm_create_ipc_workspace_for_all_reduce_fusion(0, 1, 64, 7168)
        assert (handles[0].oneshot is not None) == (flag == "1")
        ag = torch.zeros(T, D, dtype=torch.bfloat16, device=DEV)
        comm.trtllm_allgather_fusion(allgather_in=x, world_size=1, world_rank=0, hidden_dim=D, workspace_ptrs=wsp,
                                     num_token_current_rank=T, allgather_out=ag, num_token_all_group=T,
                                     pattern_code=comm.AllGatherFusionPattern.kAllGather)
        assert torch.equal(ag, x)
        ag2 = torch.zeros_like(ag)
        xn = torch.zeros(T, QR, dtype=torch.bfloat16, device=DEV)
        q = torch.zeros(T, QR, dtype=torch.float8_e4m3fn, device=DEV)
        sc = torch.zeros(T, QR // 128, device=DEV)
        comm.trtllm_allgather_fusion(allgather_in=x, world_size=1, world_rank=0, hidden_dim=D, workspace_ptrs=wsp,
                                     num_token_current_rank=T, allgather_out=ag2, num_token_all_group=T,
                                     pattern_code=comm.AllGatherFusionPattern.kAllGatherfusedRMSFP8BlockWiseQuant,
                                     x_norm_out=xn, y_norm_out=ag2, quant_out=q, scale_out=sc, x_rms_gamma=gq, y_rms_gamma=gkv,
                                     x_rms_eps=1e-6, y_rms_eps=1e-6, q_lora_rank=QR, kv_lora_rank=KVR, qk_rope_head_dim=64)
        torch.cuda.synchronize()
        got.append([ag2.clone(), xn.clone(), q.view(torch.uint8).clone(), sc.clone()])
        if handles[0].oneshot is not None:
            handles[0].oneshot.check()
        comm.trtllm_destroy_ipc_workspace_for_all_reduce_fusion(handles)
    for a, b in zip(*got):
        assert torch.equal(a, b)
    assert not torch.equal(got[0][0][:, QR:QR + KVR], x[:, QR:QR + KVR])      # the kv columns were normalised in place
    # TPDPConvertor + vllm_ar at world 1 on the one-shot transport
    monkeypatch.setenv("FLUENT_ONESHOT", "1")
    cv = TPDPConvertor(TPDPConvertor.Params(0, 64, 1, 7168), device=torch.device(DEV))
    assert cv.oneshot is not None
    rs = cv.get_reduce_scatter_context(29)
    rs.input().copy_(torch.randn(29, 7168, generator=g).to(torch.bfloat16))
    cv.reduce_scatter(rs)
    agc = cv.get_all_gather_context(29)
    agc.input().copy_(rs.output())
    cv.all_gather(agc)
    torch.cuda.synchronize()
    assert torch.equal(rs.output(), rs.input()) and torch.equal(agc.output(), rs.input())
    cv.oneshot.check()
    h = vllm_ar.init_custom_ar([], torch.empty(0), 0, True)
    y = torch.randn(17, 7168, generator=g).to(torch.bfloat16).to(DEV)
    y0 = y.clone()
    vllm_ar.all_reduce(h, y, y)
    z = torch.empty_like(y)
    vllm_ar.all_reduce(h, y, z)
    torch.cuda.synchronize()
    assert torch.equal(y, y0) and torch.equal(z, y0)
    vllm_ar.dispose(h)


def test_oneshot_beside_mla_decode_on_a_second_stream():
    """the one-shot fused all-reduce between two processes while EACH keeps a second stream busy with MLA decode launches that
    occupy every CU (8 waves x 256 VGPRs, 160 KB of LDS per workgroup): the push workgroups queue behind K1 workgroups —
    latency, never a timeout, sums bit-identical to the RCCL route's kernel"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "oneshot_two_procs.py"), "2", "k1"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("beside K1 OK") == 2, r.stdout[-2000:]


def test_oneshot_lost_peer_poisons_outputs_and_raises():
    """A peer that never issues an operation: the waiting rank's launch ends after its time budget with NaN in every output
    row (never partial sums), the epoch is not advanced, the next launch call and check() raise, and the late rank fails
    the same way instead of consuming rows of the wrong epoch."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "oneshot_two_procs.py"), "2", "lostpeer"], capture_output=True,
                       text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("lost-peer handling OK") == 2, r.stdout[-2000:]


def test_fusion_entry_points_fall_back_to_the_collective_route_beyond_the_oneshot_capacity(monkeypatch):
    """a token count above the one-shot workspace's capacity (here 16) silently takes the RCCL route (world 1: no exchange)
    with identical results; nothing raises"""
    import flashinfer.comm as comm
    monkeypatch.setenv("FLUENT_ONESHOT", "1")
    H = 2048
    handles, wsp = comm.trtllm_create_ipc_workspace_for_all_reduce_fusion(0, 1, 16, H)
    assert handles[0].oneshot is not None and handles[0].oneshot.max_tokens == 16
    g = torch.Generator().manual_seed(9)
    gamma = torch.rand(H, generator=g).to(torch.bfloat16).to(DEV)
    outs = []
    for T in (16, 17, 300):
        x = torch.randn(T, H, generator=g).to(torch.bfloat16).to(DEV)
        res = torch.randn(T, H, generator=g).to(torch.bfloat16).to(DEV)
        r, n = torch.empty_like(x), torch.empty_like(x)
        comm.trtllm_allreduce_fusion(allreduce_in=x, world_size=1, world_rank=0, token_num=T, hidden_dim=H, workspace_ptrs=wsp,
                                     pattern_code=comm.AllReduceFusionPattern.kARResidualRMSNorm, residual_in=res, residual_out=r,
                                     norm_out=n, rms_gamma=gamma, rms_eps=1e-6)
        torch.cuda.synchronize()
        x32 = x.float() + res.float()
        assert torch.equal(r, x32.to(torch.bfloat16))
        y = (x32 * torch.rsqrt(x32.pow(2).mean(-1, keepdim=True) + 1e-6)).to(torch.bfloat16) * gamma
        assert float((n.float() - y.float()).abs().max()) <= 2 ** -6 * float(y.float().abs().max())   # one bf16 ulp of slack
    handles[0].oneshot.check()
    comm.trtllm_destroy_ipc_workspace_for_all_reduce_fusion(handles)
