"""CPU, world_size 2, gloo: host logic of the comm-fused ops (flashinfer.comm.trtllm_{allreduce,reducescatter,allgather}
_fusion + all_gather.simple_all_gather; call sites flashinfer_comm_fusion.py:372-397, 485-509, 613-638, 271-283) — the
one-shot exchanges with the uneven token split of get_num_tokens_per_rank, residual scattering and output slicing.
The fused kernels are replaced by the oracle's arithmetic (tests/comm_torch_ops.py)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(T, H, world):
    g = torch.Generator().manual_seed(7)
    xs = [(torch.randn(T, H, generator=g)).to(torch.bfloat16) for _ in range(world)]
    res = torch.randn(T, H, generator=g).to(torch.bfloat16)
    add = torch.randn(T, H, generator=g).to(torch.bfloat16)
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(torch.bfloat16)
    return xs, res, add, gamma


def _worker(rank, world, port, T):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "sglang-fluentllm_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from comm_torch_ops import TorchNormOps
        from oracle import gemm_ref, norm_ref
        import flashinfer.comm as comm
        from fluent_mi355.comm import get_num_tokens_per_rank, set_norm_ops

        set_norm_ops(TorchNormOps())
        H, eps = 256, 1e-6
        xs, res, add, gamma = _inputs(T, H, world)
        pieces = torch.stack(xs)
        counts = get_num_tokens_per_rank(world, T)
        lo, hi = sum(counts[:rank]), sum(counts[:rank + 1])
        _, ws = comm.trtllm_create_ipc_workspace_for_all_reduce_fusion(rank, world, 64, H, group=None)

        # ---- C5: all-reduce + residual + RMSNorm + fp8 block quant, full residual ----
        y_ref, r_ref = norm_ref.fused_add_rmsnorm(pieces, None, res, gamma, eps)
        q_ref, s_ref = gemm_ref.per_token_group_quant_fp8(y_ref.contiguous(), 128)
        norm_out, res_out = torch.empty(T, H, dtype=torch.bfloat16), torch.empty(T, H, dtype=torch.bfloat16)
        quant_out = torch.empty(T, H, dtype=torch.float8_e4m3fn)
        scale_out = torch.empty(T, H // 128)
        comm.trtllm_allreduce_fusion(allreduce_in=xs[rank], world_size=world, world_rank=rank, token_num=T, hidden_dim=H,
                                     workspace_ptrs=ws, pattern_code=comm.AllReduceFusionPattern.kARResidualRMSNormFP8BlockWiseQuant,
                                     residual_in=res, residual_out=res_out, norm_out=norm_out, quant_out=quant_out,
                                     scale_out=scale_out, rms_gamma=gamma, rms_eps=eps)
        assert torch.equal(norm_out, y_ref) and torch.equal(res_out, r_ref)
        assert torch.equal(quant_out.view(torch.uint8), q_ref.view(torch.uint8)) and torch.equal(scale_out, s_ref)

        # ---- C5 with a reduce-scattered residual and a partial norm output (layernorm.py:114-153) ----
        res_sc_out = torch.empty(hi - lo, H, dtype=torch.bfloat16)
        partial = torch.empty(hi - lo, H, dtype=torch.bfloat16)
        norm_out2 = torch.empty(T, H, dtype=torch.bfloat16)
        comm.trtllm_allreduce_fusion(allreduce_in=xs[rank], world_size=world, world_rank=rank, token_num=T, hidden_dim=H,
                                     workspace_ptrs=ws, pattern_code=comm.AllReduceFusionPattern.kARResidualRMSNormPartialOut,
                                     residual_in=res[lo:hi].contiguous(), residual_out=res_sc_out, norm_out=norm_out2,
                                     rms_gamma=gamma, rms_eps=eps, residual_reduce_scattered=True, partial_norm_out=partial)
        # the residual slices are gathered and enter the fused kernel as its fp32 residual operand: bit-identical to the
        # full-residual call (and to forward_native)
        assert torch.equal(norm_out2, y_ref)
        assert torch.equal(partial, norm_out2[lo:hi])
        assert torch.equal(res_sc_out, r_ref[lo:hi])

        # ---- C6: reduce-scatter (uneven split) + add_in + residual + RMSNorm ----
        n = hi - lo
        y6, r6 = norm_ref.fused_add_rmsnorm(pieces[:, lo:hi], add[lo:hi], res[lo:hi], gamma, eps)
        norm6, res6 = torch.empty(n, H, dtype=torch.bfloat16), torch.empty(n, H, dtype=torch.bfloat16)
        comm.trtllm_reducescatter_fusion(reducescatter_in=xs[rank], world_size=world, world_rank=rank, token_num=T,
                                         hidden_dim=H, workspace_ptrs=ws, num_token_current_rank=n,
                                         pattern_code=comm.ReduceScatterFusionPattern.kRSAddResidualRMSNorm,
                                         add_in=add[lo:hi].contiguous(), residual_in=res[lo:hi].contiguous(), residual_out=res6,
                                         norm_out=norm6, rms_gamma=gamma, rms_eps=eps)
        assert torch.equal(norm6, y6) and torch.equal(res6, r6)

        # ---- C7: all-gather (uneven) + dual RMSNorm, y in place on the gathered tensor ----
        q_rank, kv_rank, rope = 128, 64, 32
        D = q_rank + kv_rank + rope
        g = torch.Generator().manual_seed(9)
        full = torch.randn(T, D, generator=g).to(torch.bfloat16)
        gq = (1 + 0.1 * torch.randn(q_rank, generator=g)).to(torch.bfloat16)
        gkv = (1 + 0.1 * torch.randn(kv_rank, generator=g)).to(torch.bfloat16)
        x_ref, ag_ref = norm_ref.dual_rmsnorm(full, q_rank, kv_rank, gq, gkv, 1e-6, 1e-5)
        ag = torch.empty(T, D, dtype=torch.bfloat16)
        xn = torch.empty(T, q_rank, dtype=torch.bfloat16)
        comm.trtllm_allgather_fusion(allgather_in=full[lo:hi].contiguous(), world_size=world, world_rank=rank, hidden_dim=D,
                                     workspace_ptrs=ws, num_token_current_rank=n, allgather_out=ag, num_token_all_group=T,
                                     pattern_code=comm.AllGatherFusionPattern.kAllGatherfusedRMS, x_norm_out=xn,
                                     y_norm_out=ag[..., q_rank:q_rank + kv_rank], x_rms_gamma=gq, y_rms_gamma=gkv,
                                     x_rms_eps=1e-6, y_rms_eps=1e-5, q_lora_rank=q_rank, kv_lora_rank=kv_rank,
                                     qk_rope_head_dim=rope)
        assert torch.equal(xn, x_ref) and torch.equal(ag, ag_ref)

        # ---- C3: eps.communication.TPDPConvertor, uneven reduce-scatter / all-gather (decoder_comm_manager.py:42-85) ----
        from eps.communication import TPDPConvertor
        conv = TPDPConvertor(TPDPConvertor.Params(rank, 64, world, H, None), device=torch.device("cpu"))
        rs = conv.get_reduce_scatter_context(T, 45)
        rs.input().view(dtype=torch.bfloat16).copy_(xs[rank])
        conv.reduce_scatter(rs, None)
        assert rs.output_row_offset == lo and torch.equal(rs.output(), pieces[:, lo:hi].float().sum(0).to(torch.bfloat16))
        ag_ctx = conv.get_all_gather_context(T, H, 28)
        if hi > lo:
            ag_ctx.input().copy_(res[lo:hi])
        conv.all_gather(ag_ctx, None)
        assert torch.equal(ag_ctx.output(), res)

        # ---- vocab all-gather ----
        _, wsv = comm.all_gather.create_ipc_workspace_for_allgather(rank, world, 16, 8 * world, False, group=None)
        loc = torch.full((3, 8), float(rank + 1), dtype=torch.bfloat16)
        outv = torch.empty(3, 8 * world, dtype=torch.bfloat16)
        comm.all_gather.simple_all_gather(allgather_in=loc, world_size=world, world_rank=rank, token_num=3, hidden_size=8,
                                          workspace_ptrs=wsv, max_num_tokens=16, allgather_out=outv)
        for r in range(world):
            assert torch.all(outv[:, 8 * r:8 * r + 8] == r + 1)
    finally:
        dist.destroy_process_group()


def _worker_tp_groups(rank, world, port, tp):
    """TPDPConvertor inside attention-TP groups smaller than the world (DP-attention: dp_attention.py:39-74): every DP replica
    has its OWN token count, the exchange must stay inside the replica's block of `tp` ranks."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "sglang-fluentllm_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from comm_torch_ops import TorchNormOps
        from fluent_mi355.comm import get_num_tokens_per_rank, set_norm_ops
        from eps.communication import TPDPConvertor

        set_norm_ops(TorchNormOps())
        H = 64
        block, r_in = rank // tp, rank % tp
        T = [5, 1][block]   # different token counts per replica (1 < tp: one rank of block 1 owns no rows)
        g = torch.Generator().manual_seed(100 + block)
        xs = [torch.randn(T, H, generator=g).to(torch.bfloat16) for _ in range(tp)]
        full = torch.randn(T, H, generator=g).to(torch.bfloat16)
        counts = get_num_tokens_per_rank(tp, T)
        lo, hi = sum(counts[:r_in]), sum(counts[:r_in + 1])
        conv = TPDPConvertor(TPDPConvertor.Params(rank, 64, tp, H, None), device=torch.device("cpu"))   # no group passed
        assert conv.world == tp and conv.rank == r_in
        rs = conv.get_reduce_scatter_context(T, 45)
        rs.input().copy_(xs[r_in])
        conv.reduce_scatter(rs, None)
        assert rs.output_row_offset == lo
        assert torch.equal(rs.output(), torch.stack(xs)[:, lo:hi].float().sum(0).to(torch.bfloat16))
        ag = conv.get_all_gather_context(T, H, 28)
        if hi > lo:
            ag.input().copy_(full[lo:hi])
        conv.all_gather(ag, None)
        assert torch.equal(ag.output(), full)
    finally:
        dist.destroy_process_group()


def test_tpdp_convertor_inside_attention_tp_groups_world4_tp2():
    port = _free_port()
    mp.spawn(_worker_tp_groups, args=(4, port, 2), nprocs=4, join=True)


@pytest.mark.parametrize("T", [5, 8, 1])
def test_comm_fusion_world2_gloo(T):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, T), nprocs=2, join=True)


def _oneshot_setup_worker(rank, world, port):
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "sglang-fluentllm_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fluent_mi355.oneshot import OneShotComm

        # no HIP device here: the local workspace allocation fails on every rank; the constructor must still run both of
        # its exchanges and raise on ALL ranks (a rank left waiting in a collective would hang this test)
        try:
            OneShotComm(rank, world, 64, 256)
            raise AssertionError("OneShotComm came up without a GPU")
        except RuntimeError as ex:
            assert "setup failed on rank 0" in str(ex) and "rank 1" in str(ex), str(ex)
        # one rank fails, the other is fine so far: still both raise, naming the failing rank
        def exchange_marker(obj):
            out = [None] * world
            dist.all_gather_object(out, obj)
            return out
        import fluent_mi355.oneshot as osm
        orig_check = osm.check

        def fake_check(status, what):   # rank 1 "succeeds" locally, rank 0 fails at create
            if rank == 1 and what in ("fl_comm_create", "fl_comm_local_handle"):
                return
            orig_check(status, what)
        osm.check = fake_check
        try:
            OneShotComm(rank, world, 64, 256, exchange=exchange_marker)
            raise AssertionError("must not connect")
        except RuntimeError as ex:
            assert "rank 0" in str(ex) and "rank 1:" not in str(ex), str(ex)
        finally:
            osm.check = orig_check
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_oneshot_setup_is_all_or_nothing_across_ranks_world2_gloo():
    port = _free_port()
    mp.spawn(_oneshot_setup_worker, args=(2, port), nprocs=2, join=True)
