"""Host-side mirror of `flashinfer.comm` on the decode hot path (reference fork: 3rdparty/flashinfer feature/longcat_main;
call sites python/sglang/srt/layers/flashinfer_comm_fusion.py:50-58, 86-93, 115-135, 271-283, 372-397, 485-509, 613-638).

MI355X mapping.  The reference's kernels are TRT-LLM "one-shot + Lamport flag" kernels over NVLink-mapped IPC buffers.
xGMI is a point-to-point mesh (7 links per GPU): the natural one-shot exchange is ONE RCCL collective in which every rank
receives its peers' pieces directly (all_gather / all_to_all_single with the deterministic uneven token split of
`get_num_tokens_per_rank`, flashinfer_comm_fusion.py:237-244), followed by ONE fused HIP kernel that reduces the received
pieces and applies add_in + residual + RMSNorm (+ 1x128 FP8 block quantisation) — `fl_fused_add_rmsnorm` /
`fl_dual_rmsnorm` (csrc/norm_fused.hip).  For the decode regime (token counts up to the workspace's capacity) C5 and C6
take the ONE-SHOT route instead: a peer-mapped (hipIpc) workspace per rank, created with the workspace object, into which
the peers write their rows over xGMI, and a single kernel that pushes, waits on per-row flags and applies the fused
epilogue — no RCCL call, one launch (`fluent_mi355/oneshot.py`, csrc/comm_oneshot.hip; FLUENT_ONESHOT=0 disables, the
RCCL route then serves every size).  Both routes sum the pieces in rank order and share the epilogue code: their results are
bit-identical.  Semantics that the (absent) third-party module leaves implicit are taken from the call sites and stated below.

`norm_ops` exists so that the multi-process HOST logic can be exercised on CPU tensors with the gloo backend (tests inject a
torch implementation); the product default is the HIP one and there is no automatic fallback."""
from __future__ import annotations

import ctypes
import enum
import os
from typing import Optional

import torch
import torch.distributed as dist


class AllReduceFusionPattern(enum.IntEnum):
    kAllReduce = 0
    kARResidualRMSNorm = 1
    kARResidualRMSNormFP8Quant = 2
    kARResidualRMSNormFP4Quant = 3
    kARResidualRMSNormOutFP8Quant = 4
    kARResidualRMSNormOutFP4Quant = 5
    kARResidualRMSNormFP8BlockWiseQuant = 6
    kARResidualRMSNormPartialOut = 7
    kARResidualRMSNormPartialOutFP8BlockWiseQuant = 8


class ReduceScatterFusionPattern(enum.IntEnum):
    kReduceScatter = 0
    kRSResidualRMSNorm = 1
    kRSResidualRMSNormFP8BlockWiseQuant = 2
    kRSAddResidualRMSNorm = 3
    kRSAddResidualRMSNormFP8BlockWiseQuant = 4


class AllGatherFusionPattern(enum.IntEnum):
    kAllGather = 0
    kAllGatherfusedRMS = 1
    kAllGatherfusedRMSFP8BlockWiseQuant = 2


def get_num_tokens_per_rank(world_size: int, total: int) -> list:
    """flashinfer_comm_fusion.py:237-244."""
    return [total // world_size + (1 if r < total % world_size else 0) for r in range(world_size)]


class HipNormOps:
    """Device implementation: every method is one C-ABI call on the current stream."""

    def __init__(self):
        from ._lib import check, lib, stream_ptr

        self._check, self._lib, self._stream = check, lib, stream_ptr
        vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
        lib.fl_fused_add_rmsnorm.argtypes = [vp, i32, i64, vp, vp, vp, f32, i64, i32, vp, vp, vp, vp, i64, i64, vp]
        lib.fl_fused_add_rmsnorm.restype = i32
        lib.fl_fused_add_rmsnorm_offset.argtypes = [vp, i32, i64, vp, vp, vp, f32, f32, i64, i32, vp, vp, vp]
        lib.fl_fused_add_rmsnorm_offset.restype = i32
        lib.fl_dual_rmsnorm.argtypes = [vp, i64, i32, i32, i32, vp, vp, f32, f32, vp, vp, vp, i64, i64, vp]
        lib.fl_dual_rmsnorm.restype = i32

    @staticmethod
    def _p(t):
        return None if t is None else t.data_ptr()

    def add_rmsnorm(self, pieces, add_in, residual_in, gamma, eps, residual_out, norm_out, quant_out, scale_out, gamma_offset=0.0):
        """pieces [W, T, H] bf16 contiguous.  gamma_offset != 0: norm_out uses (gamma + gamma_offset), added in fp32 (GemmaRMSNorm)."""
        W, T, H = pieces.shape
        if T == 0:
            return   # a rank without token rows (T < world): nothing to compute; the collectives around this call still ran
        for t in (pieces, add_in, residual_in, gamma, residual_out, norm_out):
            assert t is None or (t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous()), "bf16 contiguous CUDA tensors"
        if gamma_offset != 0.0:
            assert quant_out is None and scale_out is None, "the offset form has no fused quantisation"
            self._check(self._lib.fl_fused_add_rmsnorm_offset(pieces.data_ptr(), W, T * H, self._p(add_in), self._p(residual_in),
                                                              self._p(gamma), float(gamma_offset), float(eps), T, H,
                                                              self._p(residual_out), self._p(norm_out), self._stream(pieces.device)),
                        "fl_fused_add_rmsnorm_offset")
            return
        sst, ssg = (scale_out.stride(0), scale_out.stride(1)) if scale_out is not None else (0, 0)
        self._check(self._lib.fl_fused_add_rmsnorm(pieces.data_ptr(), W, T * H, self._p(add_in), self._p(residual_in),
                                                   self._p(gamma), float(eps), T, H, self._p(residual_out),
                                                   self._p(norm_out), self._p(quant_out), self._p(scale_out), sst, ssg,
                                                   self._stream(pieces.device)), "fl_fused_add_rmsnorm")

    def dual_rmsnorm(self, ag, q_rank, kv_rank, gamma_q, gamma_kv, eps_q, eps_kv, x_norm_out, quant_out, scale_out):
        T, D = ag.shape
        if T == 0:
            return
        assert ag.is_cuda and ag.dtype == torch.bfloat16 and ag.is_contiguous()
        sst, ssg = (scale_out.stride(0), scale_out.stride(1)) if scale_out is not None else (0, 0)
        self._check(self._lib.fl_dual_rmsnorm(ag.data_ptr(), T, D, q_rank, kv_rank, gamma_q.data_ptr(), gamma_kv.data_ptr(),
                                              float(eps_q), float(eps_kv), self._p(x_norm_out), self._p(quant_out),
                                              self._p(scale_out), sst, ssg, self._stream(ag.device)), "fl_dual_rmsnorm")


_norm_ops = None


def set_norm_ops(ops):
    """Tests only: inject an implementation of the two fused kernels (CPU tensors + gloo)."""
    global _norm_ops
    _norm_ops = ops


def _ops():
    global _norm_ops
    if _norm_ops is None:
        _norm_ops = HipNormOps()
    return _norm_ops


class _Workspace:
    """What `workspace_ptrs` resolves to: the process group of the RCCL route and, when it could be set up, the one-shot
    peer-mapped communicator (`oneshot`, fluent_mi355/oneshot.py)."""

    def __init__(self, rank, world_size, group):
        self.rank, self.world_size, self.group = rank, world_size, group
        self.oneshot = None


_registry = {}


def _register(rank, world_size, group, device=None):
    ws = _Workspace(rank, world_size, group)
    tensor = torch.zeros(1, dtype=torch.int64, device=device if device is not None else "cpu")
    _registry[id(tensor)] = ws
    tensor._fluent_ws = ws
    return [ws], tensor


def _resolve(workspace_ptrs) -> _Workspace:
    ws = getattr(workspace_ptrs, "_fluent_ws", None)
    if ws is None:
        ws = _registry.get(id(workspace_ptrs))
    if ws is None:
        raise RuntimeError("workspace_ptrs was not created by trtllm_create_ipc_workspace_for_all_reduce_fusion")
    return ws


def _device_of_group(group):
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() and (
        group is None or dist.get_backend(group) == "nccl") else None


def trtllm_create_ipc_workspace_for_all_reduce_fusion(rank, world_size, max_token_num, hidden_dim, group=None,
                                                      use_fp32_lamport=False):
    """flashinfer_comm_fusion.py:86-93 -> (ipc_handles, workspace_tensor).  With a HIP device and an RCCL group of more than
    one rank this also allocates the rank's one-shot workspace and maps the peers' (a collective call, like the
    reference's: every rank of the group must make it).  FLUENT_ONESHOT=0: never; =1: also at world 1 (tests)."""
    handles, tensor = _register(rank, world_size, group, _device_of_group(group) if dist.is_initialized() else None)
    want = os.environ.get("FLUENT_ONESHOT", "auto")
    multi = dist.is_initialized() and world_size > 1 and dist.get_world_size(group) == world_size and dist.get_backend(group) == "nccl"
    if want != "0" and torch.cuda.is_available() and (multi or want == "1"):
        from .oneshot import OneShotComm

        # OneShotComm's constructor is collective and all-or-nothing: either every rank of the group is connected or every
        # rank gets the RuntimeError (hipIpc unavailable between two devices, ...) — then the RCCL route serves every size
        comm = None
        try:
            comm = OneShotComm(rank if multi else 0, world_size if multi else 1, max_token_num, hidden_dim, group=group)
        except RuntimeError as ex:
            import warnings

            warnings.warn(f"fluent_mi355: one-shot peer-mapped C5/C6 unavailable ({ex}); using the RCCL route for every size")
        handles[0].oneshot = comm
    return handles, tensor


def trtllm_destroy_ipc_workspace_for_all_reduce_fusion(ipc_handles, group=None):
    """flashinfer_comm_fusion.py:115-117."""
    for ws in ipc_handles or []:
        if getattr(ws, "oneshot", None) is not None:
            ws.oneshot.close()
            ws.oneshot = None
    return None


def destroy_ipc_workspace_for_allgather(ipc_handles, group=None):
    """flashinfer_comm_fusion.py:133-135."""
    return None


def _world(ws):
    return ws.world_size if (dist.is_initialized() and ws.world_size > 1) else 1


def _sum_pieces(pieces, out):
    """out = sum over the W pieces (fp32 accumulate) through the fused kernel: no eager torch arithmetic on this path."""
    if out.dtype == pieces.dtype and out.is_contiguous():
        _ops().add_rmsnorm(pieces.contiguous(), None, None, None, 0.0, out, None, None, None)
    else:
        tmp = torch.empty(pieces.shape[1:], dtype=pieces.dtype, device=pieces.device)
        _ops().add_rmsnorm(pieces.contiguous(), None, None, None, 0.0, tmp, None, None, None)
        out.copy_(tmp)


_compact_idx = {}


def _gather_rows_uneven(rows, counts, rank, total, group, out=None):
    """every rank contributes `rows` [counts[rank], H] -> [total, H] on every rank: ONE all-gather (RCCL sends every row
    once per peer; nothing is replicated locally).  Counts from `get_num_tokens_per_rank` differ by at most one row: the
    uneven case gathers slices padded to the largest count and compacts them with one indexed copy."""
    W = len(counts)
    if out is None:
        out = torch.empty((total,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    cmax = max(counts)
    if all(c == cmax for c in counts) and out.is_contiguous():
        dist.all_gather_into_tensor(out, rows.contiguous(), group=group)
        return out
    padded = rows.new_zeros((cmax,) + tuple(rows.shape[1:]))
    padded[:counts[rank]].copy_(rows)
    buf = torch.empty((W * cmax,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    torch.index_select(buf, 0, _compaction_index(counts, rows.device), out=out)
    return out


def _compaction_index(counts, device):
    """row t of the compacted [total] layout -> its row in the padded [W * cmax] gather.  Built ON the device from arange
    arithmetic (no pageable host-to-device copy: safe to meet a new `counts` for the first time inside hipGraph capture,
    and no host sync outside it) and cached per (counts, device).  get_num_tokens_per_rank: the first `rem` ranks own
    `base + 1` rows, the rest `base`."""
    key = (tuple(counts), str(device))
    idx = _compact_idx.get(key)
    if idx is None:
        W, total, cmax = len(counts), sum(counts), max(counts)
        base, rem = total // W, total % W
        assert list(counts) == [base + (1 if r < rem else 0) for r in range(W)], "counts do not follow get_num_tokens_per_rank"
        t = torch.arange(total, dtype=torch.int64, device=device)
        big = rem * (base + 1)
        r_hi = t // (base + 1)                                   # owner if t < big
        r_lo = rem + (t - big) // max(base, 1)                   # owner otherwise
        owner = torch.where(t < big, r_hi, r_lo)
        start = owner * base + torch.clamp(owner, max=rem)       # first row of the owner's slice
        idx = owner * cmax + (t - start)
        _compact_idx[key] = idx
    return idx


def _gather_pieces(x, ws):
    """[T, H] on every rank -> [W, T, H]: the one-shot all-reduce exchange (every rank receives every peer's tensor)."""
    W = _world(ws)
    if W == 1:
        return x.unsqueeze(0)
    buf = torch.empty((W * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(buf, x.contiguous(), group=ws.group)
    return buf.view((W,) + tuple(x.shape))


def trtllm_allreduce_fusion(allreduce_in, world_size, world_rank, token_num, hidden_dim, workspace_ptrs, launch_with_pdl=True,
                            use_oneshot=None, trigger_completion_at_end=False, fp32_acc=False, pattern_code=None,
                            allreduce_out=None, residual_in=None, residual_out=None, norm_out=None, quant_out=None,
                            scale_out=None, rms_gamma=None, rms_eps=1e-6, scale_factor=None, layout_code=None,
                            residual_reduce_scattered=False, max_sm_to_use=None, partial_norm_out=None):
    """C5 (flashinfer_comm_fusion.py:372-397).  sum over ranks of `allreduce_in` [T, H]; + residual; RMSNorm; optional 1x128
    FP8 quantisation of the norm.  `residual_reduce_scattered`: residual_in / residual_out hold only this rank's token slice
    (`get_num_tokens_per_rank`) — every rank then adds its residual slice into its own piece before the exchange (the sum
    contains each residual row exactly once) and keeps its slice of the new residual.  `partial_norm_out`: this rank's token
    slice of the norm (layernorm.py:114-153 has_partial_norm_out)."""
    ws = _resolve(workspace_ptrs)
    W = _world(ws)
    T, H = allreduce_in.shape
    only_sum = pattern_code is not None and int(pattern_code) == int(AllReduceFusionPattern.kAllReduce)
    osc = ws.oneshot
    if (osc is not None and use_oneshot is not False and osc.world == W and osc.fits(T, H) and not residual_reduce_scattered
            and partial_norm_out is None and (only_sum or allreduce_out is None)):
        # one launch: push over xGMI, per-row flags, fused epilogue (csrc/comm_oneshot.hip)
        x = allreduce_in.contiguous()
        if only_sum:
            if osc.accepts(x, False, residual_out=allreduce_out):
                osc.allreduce_fused(x, residual_out=allreduce_out)
                return
        elif osc.accepts(x, False, None, residual_in, rms_gamma, residual_out, norm_out, quant_out, scale_out):
            osc.allreduce_fused(x, residual_in, rms_gamma, rms_eps, residual_out, norm_out, quant_out, scale_out)
            return
        # (tensors the one-shot kernel's raw-pointer contract does not cover — other dtypes, strided views: RCCL route)
    counts = get_num_tokens_per_rank(W, T)
    lo = sum(counts[:world_rank]) if W > 1 else 0
    hi = lo + (counts[world_rank] if W > 1 else T)
    res_full, res_out_full = residual_in, residual_out
    if residual_reduce_scattered and W > 1:
        # the residual slices are gathered (one small uneven exchange) and enter the fused kernel as its fp32 residual
        # operand: adding a slice into the bf16 input before the exchange would round (x + residual) once more than the
        # reference's fused kernel / forward_native do
        res_full = _gather_rows_uneven(residual_in, counts, world_rank, T, ws.group) if residual_in is not None else None
        res_out_full = torch.empty_like(allreduce_in) if residual_out is not None else None
    pieces = _gather_pieces(allreduce_in, ws)
    want_norm = pattern_code is None or int(pattern_code) != int(AllReduceFusionPattern.kAllReduce)
    if want_norm:
        n_out = norm_out if norm_out is not None else (torch.empty_like(allreduce_in) if partial_norm_out is not None else None)
        _ops().add_rmsnorm(pieces.contiguous(), None, res_full, rms_gamma, rms_eps, res_out_full, n_out, quant_out, scale_out)
        if residual_reduce_scattered and W > 1 and residual_out is not None:
            residual_out.copy_(res_out_full[lo:hi])
        if partial_norm_out is not None:
            partial_norm_out.copy_(n_out[lo:hi])
    if allreduce_out is not None:
        _sum_pieces(pieces, allreduce_out)


def trtllm_reducescatter_fusion(reducescatter_in, world_size, world_rank, token_num, hidden_dim, workspace_ptrs,
                                launch_with_pdl=True, trigger_completion_at_end=False, num_token_current_rank=None,
                                fp32_acc=False, pattern_code=None, use_oneshot=None, reducescatter_out=None, add_in=None,
                                residual_in=None, residual_out=None, norm_out=None, quant_out=None, scale_out=None,
                                rms_gamma=None, rms_eps=1e-6, scale_factor=None, layout_code=None):
    """C6 (flashinfer_comm_fusion.py:485-509).  This rank's token slice of the sum over ranks of `reducescatter_in` [T, H]
    (+ add_in) + residual; RMSNorm; optional FP8 quantisation.  All per-rank tensors are [num_token_current_rank, H]."""
    ws = _resolve(workspace_ptrs)
    W = _world(ws)
    T, H = reducescatter_in.shape
    only_sum = pattern_code is not None and int(pattern_code) == int(ReduceScatterFusionPattern.kReduceScatter)
    osc = ws.oneshot
    if (osc is not None and use_oneshot is not False and osc.world == W and osc.fits(T, H, reduce_scatter=True)
            and (only_sum or reducescatter_out is None)):
        x = reducescatter_in.contiguous()
        if only_sum:
            if osc.accepts(x, True, residual_out=reducescatter_out):
                osc.reducescatter_fused(x, residual_out=reducescatter_out)
                return
        elif osc.accepts(x, True, add_in, residual_in, rms_gamma, residual_out, norm_out, quant_out, scale_out):
            osc.reducescatter_fused(x, add_in, residual_in, rms_gamma, rms_eps, residual_out, norm_out, quant_out, scale_out)
            return
    counts = get_num_tokens_per_rank(W, T)
    mine = counts[world_rank] if W > 1 else T
    if W == 1:
        pieces = reducescatter_in.unsqueeze(0)
    else:
        # one-shot reduce-scatter: every rank receives its token slice from every peer (uneven splits), reduces locally
        recv = torch.empty((W * mine, H), dtype=reducescatter_in.dtype, device=reducescatter_in.device)
        dist.all_to_all_single(recv, reducescatter_in.contiguous(), output_split_sizes=[mine] * W,
                               input_split_sizes=counts, group=ws.group)
        pieces = recv.view(W, mine, H)
    if pattern_code is not None and int(pattern_code) == int(ReduceScatterFusionPattern.kReduceScatter):
        _sum_pieces(pieces, reducescatter_out)
        return
    _ops().add_rmsnorm(pieces.contiguous(), add_in, residual_in, rms_gamma, rms_eps, residual_out, norm_out, quant_out, scale_out)
    if reducescatter_out is not None:
        _sum_pieces(pieces, reducescatter_out)


def trtllm_allgather_fusion(allgather_in, world_size, world_rank, hidden_dim, workspace_ptrs, launch_with_pdl=True,
                            trigger_completion_at_end=False, num_token_current_rank=None, allgather_out=None,
                            num_token_all_group=None, pattern_code=None, use_oneshot=True, fp32_acc=False, x_norm_out=None,
                            y_norm_out=None, quant_out=None, scale_out=None, x_rms_gamma=None, y_rms_gamma=None,
                            x_rms_eps=1e-6, y_rms_eps=1e-6, q_lora_rank=None, kv_lora_rank=None, qk_rope_head_dim=None):
    """C7 (flashinfer_comm_fusion.py:613-638).  Gathers the token rows of every rank (uneven counts =
    `get_num_tokens_per_rank(world, num_token_all_group)`) into allgather_out [total, D], then RMSNorm of the q_a columns
    into x_norm_out (+ optional FP8 quantisation) and of the kv_a columns IN PLACE (y_norm_out aliases allgather_out,
    flashinfer_comm_fusion.py:578-579)."""
    ws = _resolve(workspace_ptrs)
    W = _world(ws)
    t_cur, D = allgather_in.shape
    osc = ws.oneshot
    only_gather = pattern_code is not None and int(pattern_code) == int(AllGatherFusionPattern.kAllGather)
    total = int(num_token_all_group) if num_token_all_group is not None else t_cur * W
    if osc is not None and use_oneshot is not False and osc.world == W:
        # one launch: peer-mapped push of every rank's rows + (C7) the dual RMSNorm on each gathered row
        x = allgather_in.contiguous()
        qr = 0 if only_gather else int(q_lora_rank)
        if osc.accepts_gather(x, total, allgather_out, None if only_gather else x_norm_out, None if only_gather else quant_out,
                              None if only_gather else scale_out, qr, 0 if only_gather else int(kv_lora_rank), x_rms_gamma, y_rms_gamma):
            osc.allgather_fused(x, total, allgather_out, qr, 0 if only_gather else int(kv_lora_rank), x_rms_gamma, y_rms_gamma,
                                x_rms_eps, y_rms_eps, None if only_gather else x_norm_out, None if only_gather else quant_out,
                                None if only_gather else scale_out)
            return
    if W == 1:
        allgather_out[:t_cur].copy_(allgather_in)
    else:
        counts = get_num_tokens_per_rank(W, num_token_all_group)
        assert counts[world_rank] == t_cur, "token split does not follow get_num_tokens_per_rank"
        _gather_rows_uneven(allgather_in, counts, world_rank, num_token_all_group, ws.group, out=allgather_out[:num_token_all_group])
    if pattern_code is not None and int(pattern_code) == int(AllGatherFusionPattern.kAllGather):
        return
    _ops().dual_rmsnorm(allgather_out, q_lora_rank, kv_lora_rank, x_rms_gamma, y_rms_gamma, x_rms_eps, y_rms_eps, x_norm_out,
                        quant_out, scale_out)


# ---- flashinfer.comm.all_gather (vocab gather, flashinfer_comm_fusion.py:50-58, 271-283) ----
def create_ipc_workspace_for_allgather(rank, world_size, max_token_num, hidden_size, use_fp32=False, group=None):
    return _register(rank, world_size, group, _device_of_group(group) if dist.is_initialized() else None)


def simple_all_gather(allgather_in, world_size, world_rank, token_num, hidden_size, workspace_ptrs, launch_with_pdl=True,
                      trigger_completion_at_end=False, max_num_tokens=None, allgather_out=None, max_sm_to_use=None):
    """[T, V_local] on every rank -> allgather_out [T, world * V_local] (rank r's columns at [r*V_local, (r+1)*V_local))."""
    ws = _resolve(workspace_ptrs)
    W = _world(ws)
    if W == 1:
        allgather_out.copy_(allgather_in)
        return allgather_out
    T = allgather_in.shape[0]
    buf = torch.empty((W * T, hidden_size), dtype=allgather_in.dtype, device=allgather_in.device)
    dist.all_gather_into_tensor(buf, allgather_in.contiguous(), group=ws.group)
    allgather_out.view(T, W, hidden_size).copy_(buf.view(W, T, hidden_size).permute(1, 0, 2))
    return allgather_out


# ---- eps.communication.TPDPConvertor (C3): uneven reduce-scatter / all-gather inside the attention-TP group
#      (layers/dp_attention.py:62-74; used through distributed/decoder_comm_manager.py:42-85 `RSAG`) ----
class _RSContext:
    def __init__(self, inp, out, offset):
        self._inp, self._out, self.output_row_offset = inp, out, offset

    def input(self):
        return self._inp

    def output(self):
        return self._out


class TPDPConvertor:
    """Token rows are split over the group as `get_token_dist` does in the reference's in-tree twin of this module
    (device_communicators/custom_triton_rsag/triton_rsag.py:48-57): rank r owns counts[r] = T//W + (r < T%W) rows at offset
    sum(counts[:r]).  reduce_scatter: every rank receives its row slice from every peer in ONE uneven all_to_all_single
    (direct over the xGMI mesh) and sums the W pieces in one kernel; all_gather: ONE uneven all_to_all_single."""

    class Params:
        def __init__(self, global_rank, max_num_tokens, tp_size, hidden_size, communicator=None):
            self.global_rank, self.max_num_tokens, self.tp_size = global_rank, max_num_tokens, tp_size
            self.hidden_size, self.communicator = hidden_size, communicator

    _groups = {}   # (world, tp_size) -> list of process groups, one per block of tp_size consecutive ranks

    @classmethod
    def _tp_group(cls, global_rank, tp_size):
        """The attention-TP group of `global_rank`: the block of `tp_size` consecutive ranks it lies in (the layout of
        dp_attention.py:39-58).  dist.new_group is collective over the WORLD: every rank creates every block, in order,
        once per (world, tp_size)."""
        world = dist.get_world_size()
        if tp_size >= world:
            if tp_size != world:
                raise RuntimeError(f"TPDPConvertor: tp_size {tp_size} exceeds the world size {world}")
            return None
        if world % tp_size:
            raise RuntimeError(f"TPDPConvertor: world size {world} is not a multiple of tp_size {tp_size}")
        key = (world, tp_size)
        if key not in cls._groups:
            cls._groups[key] = [dist.new_group(list(range(b * tp_size, (b + 1) * tp_size))) for b in range(world // tp_size)]
        return cls._groups[key][global_rank // tp_size]

    def __init__(self, params, group=None, device=None, dtype=torch.bfloat16):
        # the reference builds it as TPDPConvertor(Params(global_rank, max_tokens, attn_tp_size, hidden, comm)) with no
        # group (dp_attention.py:62-74): the exchange runs INSIDE the attention-TP group derived from the params
        if group is None and dist.is_initialized() and params.tp_size is not None:
            group = self._tp_group(params.global_rank if params.global_rank is not None else dist.get_rank(), int(params.tp_size))
        self.p, self.group, self.dtype = params, group, dtype
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = device if device is not None else (_device_of_group(group) if dist.is_initialized() else None) or (
            torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        # decode-sized exchanges (<= 1024 rows) go through the peer-mapped one-shot transport of csrc/comm_oneshot.hip — one
        # launch, no RCCL call — like C5/C6; larger ones keep the RCCL route below.  The constructor of OneShotComm is
        # collective over the group and all-or-nothing (every rank of the reference builds its TPDPConvertor at the same
        # point of start-up, dp_attention.py:97-131).  FLUENT_ONESHOT=0 disables, =1 also builds it at world 1 (tests).
        self.oneshot = None
        want = os.environ.get("FLUENT_ONESHOT", "auto")
        multi = dist.is_initialized() and self.world > 1 and dist.get_backend(group) == "nccl"
        if want != "0" and self.device.type == "cuda" and dtype == torch.bfloat16 and (multi or want == "1") \
                and params.hidden_size % 8 == 0 and params.hidden_size <= 8192:
            from .oneshot import MAX_ONESHOT_TOKENS, OneShotComm
            try:
                # the convertor only reduce-scatters and all-gathers: a source sends at most ceil(T / world) rows per peer, so the
                # inbox is sized for that (at world 8 and hidden 7168: 29 MB instead of 235 MB of uncached memory)
                w_ = self.world if multi else 1
                cap = max(1, min(MAX_ONESHOT_TOKENS, int(params.max_num_tokens or MAX_ONESHOT_TOKENS)))
                self.oneshot = OneShotComm(self.rank if multi else 0, w_, (cap + w_ - 1) // w_, params.hidden_size, group=group)
            except RuntimeError as ex:
                import warnings

                warnings.warn(f"fluent_mi355: one-shot TPDPConvertor unavailable ({ex}); using the RCCL route for every size")

    def get_token_dist(self, total):
        return get_num_tokens_per_rank(self.world, total)

    def get_reduce_scatter_context(self, tp_num_tokens, num_blocks=None):
        counts = self.get_token_dist(tp_num_tokens)
        inp = torch.empty(tp_num_tokens, self.p.hidden_size, dtype=self.dtype, device=self.device)
        out = torch.empty(counts[self.rank], self.p.hidden_size, dtype=self.dtype, device=self.device)
        ctx = _RSContext(inp, out, sum(counts[:self.rank]))
        ctx.counts = counts
        return ctx

    def reduce_scatter(self, ctx, stream=None):
        W, mine, H = self.world, ctx.counts[self.rank], self.p.hidden_size
        osc = self.oneshot
        if osc is not None and osc.world == W and osc.fits(ctx.input().shape[0], H, reduce_scatter=True) \
                and osc.accepts(ctx.input(), True, residual_out=ctx.output()):
            osc.reducescatter_fused(ctx.input(), residual_out=ctx.output())   # sum of the W pieces of this rank's slice, one launch
            return
        if W == 1:
            ctx.output().copy_(ctx.input())
            return
        recv = torch.empty(W * mine, H, dtype=self.dtype, device=self.device)
        dist.all_to_all_single(recv, ctx.input(), output_split_sizes=[mine] * W, input_split_sizes=ctx.counts, group=self.group)
        _ops().add_rmsnorm(recv.view(W, mine, H), None, None, None, 0.0, ctx.output(), None, None, None)

    def get_all_gather_context(self, tp_num_tokens, hidden_size=None, num_blocks=None):
        H = hidden_size if hidden_size is not None else self.p.hidden_size
        counts = self.get_token_dist(tp_num_tokens)
        inp = torch.empty(counts[self.rank], H, dtype=self.dtype, device=self.device)
        out = torch.empty(tp_num_tokens, H, dtype=self.dtype, device=self.device)
        ctx = _RSContext(inp, out, sum(counts[:self.rank]))
        ctx.counts = counts
        return ctx

    def all_gather(self, ctx, stream=None):
        W, mine = self.world, ctx.counts[self.rank]
        osc = self.oneshot
        if osc is not None and osc.world == W and osc.accepts_gather(ctx.input(), sum(ctx.counts), ctx.output()):
            osc.allgather_fused(ctx.input(), sum(ctx.counts), ctx.output())
            return
        if W == 1:
            ctx.output().copy_(ctx.input())
            return
        _gather_rows_uneven(ctx.input(), ctx.counts, self.rank, sum(ctx.counts), self.group, out=ctx.output())


# ---- eps.communication.MscclppCommunicator / MscclppCommunicatorParams (srt/distributed/parallel_state.py:52,963-977) ----
# The reference builds ONE communicator per process at start-up — rank 0 draws `createUniqueId()`, broadcasts it over the CPU group, every
# rank constructs `MscclppCommunicator(unique_id, MscclppCommunicatorParams(rank, world_size, num_ranks_per_node))` — and hands it on as
# `comm.data_ptr()` to eps.fast_ep.AllToAll (moe/dispatcher/fast_ep.py:15-22) and as the object itself to TPDPConvertor.Params
# (dp_attention.py:64-72).  MI355X: the transport is torch.distributed over RCCL / xGMI (+ the peer-mapped one-shot kernels), so the
# communicator is a HOST object that names the process group the exchanges run on; `data_ptr()` is a key into a process-local registry
# from which AllToAll recovers the object.
_communicators = {}


class MscclppCommunicatorParams:
    def __init__(self, rank, world_size, num_ranks_per_node):
        self.rank, self.world_size, self.num_ranks_per_node = int(rank), int(world_size), int(num_ranks_per_node)
        if not (0 <= self.rank < self.world_size) or self.num_ranks_per_node < 1:
            raise ValueError(f"MscclppCommunicatorParams: rank {rank} / world_size {world_size} / num_ranks_per_node {num_ranks_per_node}")


class MscclppCommunicator:
    @staticmethod
    def createUniqueId():
        """A picklable token (rank 0 broadcasts it with broadcast_object_list): 16 random bytes as a hex string."""
        return os.urandom(16).hex()

    def __init__(self, unique_id, params, group=None):
        if not isinstance(unique_id, str) or len(unique_id) != 32:
            raise ValueError("MscclppCommunicator: unique_id must come from MscclppCommunicator.createUniqueId()")
        self.unique_id, self.params = unique_id, params
        # the world group of torch.distributed (the reference's `_WORLD`); resolved lazily so that a communicator may be built first
        self._group = group
        if dist.is_initialized():
            if dist.get_world_size(group) != params.world_size or dist.get_rank(group) != params.rank:
                raise RuntimeError(f"MscclppCommunicator: params (rank {params.rank} of {params.world_size}) do not match the process "
                                   f"group (rank {dist.get_rank(group)} of {dist.get_world_size(group)})")
        elif params.world_size != 1:
            raise RuntimeError("MscclppCommunicator: torch.distributed must be initialised before a multi-rank communicator is built")
        self._key = id(self)
        _communicators[self._key] = self

    @property
    def group(self):
        return self._group

    @property
    def rank(self):
        return self.params.rank

    @property
    def world_size(self):
        return self.params.world_size

    def data_ptr(self):
        return self._key

    def close(self):
        _communicators.pop(self._key, None)


def communicator_from_ptr(comm_ptr):
    """The MscclppCommunicator behind a `data_ptr()` value (None for the values tests / tools pass: None, 0)."""
    if isinstance(comm_ptr, MscclppCommunicator):
        return comm_ptr
    if not comm_ptr:
        return None
    c = _communicators.get(int(comm_ptr))
    if c is None:
        raise RuntimeError(f"comm_ptr {comm_ptr} does not name a live MscclppCommunicator of this process")
    return c
