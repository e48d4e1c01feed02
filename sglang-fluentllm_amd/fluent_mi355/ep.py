"""Host-side mirror of `eps.fast_ep.AllToAll` (reference: 3rdparty/eps over MSCCL++; call sites
python/sglang/srt/layers/moe/dispatcher/fast_ep.py:16-22,45-51,73-78).

MI355X mapping: xGMI is a point-to-point mesh, so the natural collective is ONE equal-split all-to-all per direction
over RCCL (`torch.distributed.all_to_all_single`, backend "nccl" = RCCL) on fixed-capacity peer slabs — no counts
exchange, no host sync, static shapes (hipGraph-friendly).

Slab sizing: a token travels to a rank ONCE, however many of its top-k experts live there (the receiving rank replicates
the row to its experts locally; on the way back it returns ONE row per token, the weighted sum over its local experts).
A peer slab therefore holds `max_tokens_per_rank` rows — the true worst case of that scheme, reached only if every token
of a rank routes to the same peer — instead of `max_tokens_per_rank * top_k` rows for one row per (token, expert) pair:
8x fewer xGMI bytes at top-8, and a slab cannot overflow (a token occupies at most one row of it).  Messages (round 3):
  dispatch: ONE all_to_all_single of rows [world*cap, hidden | top_k int32 local expert ids | top_k f32 weights] — the ids (and the
            routing weights, when `dispatch(..., weights=)` is given them) travel in the TAIL of their slab row (+64 B on 14 KiB)
  combine : rows [world*cap, hidden] back — ONE all_to_all_single when the weights went out with the dispatch; otherwise (the
            reference's call order: `AllToAll.dispatch` is not given the weights, fast_ep.py:45-51) the weights follow as a second,
            32 B/row message at combine time, as before
Transport (round 4): with RCCL ("nccl") groups on the GPU and decode-sized slabs (world * cap <= 1024 rows) each of these messages is
ONE launch of the peer-mapped one-shot transport (csrc/comm_oneshot.hip: oneshot_a2a_kernel — rows pushed straight into the peers'
inboxes over their xGMI links, per-row flags, empty slab rows travel as their 64-byte tail only) and no RCCL call, like C3-C7; larger
slabs, other backends and FLUENT_ONESHOT=0 keep the RCCL all_to_all_single.  `comm_route` names the route taken.
The integer / row work around the exchanges runs in HIP kernels (csrc/ep_a2a.hip).  `row_ops` exists so that the
multi-process HOST logic can be exercised on CPU tensors with the gloo backend in tests (tests/ inject a torch-indexing
implementation); the product default is the HIP one and there is no automatic fallback."""
from __future__ import annotations

import torch
import torch.distributed as dist


class HipRowOps:
    """Device implementation: every method is one C-ABI call on the current stream."""

    def __init__(self):
        import ctypes

        from ._lib import check, lib, stream_ptr

        self._ct, self._check, self._lib, self._stream = ctypes, check, lib, stream_ptr
        vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        lib.fl_ep_route.argtypes = [vp, i64, i32, i32, i32, vp, vp, vp]
        lib.fl_ep_route_dedup.argtypes = [vp, i64, i32, i32, i32, i32, vp, vp, vp, i64, vp, vp, i64, vp]
        lib.fl_ep_sort.argtypes = [vp, i64, i32, vp, vp, vp, i32, i64, vp]
        lib.fl_ep_gather_rows.argtypes = [vp, i64, vp, i64, i32, vp, i64, vp]
        lib.fl_ep_gather_rows_div.argtypes = [vp, i64, vp, i64, i32, i32, vp, i64, vp, i64, vp]
        lib.fl_ep_scatter_rows.argtypes = [vp, i64, vp, i64, i32, vp, i64, vp]
        lib.fl_ep_send_rows.argtypes = [vp, i64, vp, i64, i32, i32, vp, i64, i64, vp]
        lib.fl_ep_combine.argtypes = [vp, i64, vp, vp, i64, i32, i32, vp, i64, vp]
        lib.fl_ep_gather_f32.argtypes = [vp, i64, vp, vp, i64, i32, i64, vp]
        for n in ("fl_ep_route", "fl_ep_route_dedup", "fl_ep_sort", "fl_ep_gather_rows", "fl_ep_gather_rows_div",
                  "fl_ep_scatter_rows", "fl_ep_send_rows", "fl_ep_combine", "fl_ep_gather_f32"):
            getattr(lib, n).restype = i32

    @staticmethod
    def _stream_of(t):
        from ._lib import stream_ptr

        if not t.is_cuda:
            raise RuntimeError("expected a CUDA/HIP tensor")
        return stream_ptr(t.device)

    # Row tensors may be VIEWS into a wider message row ([rows, cols] with stride(1) == 1 and any row stride): the kernels take
    # the row stride in the view's own element type.
    @staticmethod
    def _rows2d(t, name):
        if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
            raise RuntimeError(f"{name} must be a [rows, cols] tensor or view with a contiguous last dimension")
        return t.stride(0) if t.shape[0] > 1 else t.shape[1]

    def route_dedup(self, indices, top_k, experts_per_rank, world, cap, tok_slot, send_eid, pair_src, weights=None, send_w=None):
        """send_eid: int32 [world*cap, top_k] (a view into the message tail is fine); weights f32 [tokens*top_k] + send_w f32
        [world*cap, top_k] (view ok): the routing weights placed beside the ids in the same launch"""
        self._check(self._lib.fl_ep_route_dedup(indices.data_ptr(), indices.numel() // top_k, top_k, experts_per_rank, world, cap,
                                                tok_slot.data_ptr(), send_eid.data_ptr(), pair_src.data_ptr(),
                                                self._rows2d(send_eid, "send_eid"),
                                                None if weights is None else weights.data_ptr(),
                                                None if send_w is None else send_w.data_ptr(),
                                                0 if send_w is None else self._rows2d(send_w, "send_w"),
                                                self._stream(send_eid.device)), "fl_ep_route_dedup")

    def sort(self, recv_eid, num_local_experts, order, exclusive_sum, inverse=None):
        """recv_eid: int32 [rows, top_k] (view ok); order / inverse index the flattened (row, j) pairs"""
        self._check(self._lib.fl_ep_sort(recv_eid.data_ptr(), recv_eid.shape[0] * recv_eid.shape[1], num_local_experts, order.data_ptr(),
                                         exclusive_sum.data_ptr(), None if inverse is None else inverse.data_ptr(), recv_eid.shape[1],
                                         self._rows2d(recv_eid, "recv_eid"), self._stream(recv_eid.device)), "fl_ep_sort")

    def gather_div(self, src, idx, n, div, dst, n_valid=None):
        """dst[i] = src[idx[i] // div] for i < min(n, n_valid[0]) (n_valid: optional int32 device scalar); src [rows, hidden] (view ok)"""
        self._check(self._lib.fl_ep_gather_rows_div(src.data_ptr(), src.shape[0], idx.data_ptr(), n, div, src.shape[1],
                                                    dst.data_ptr(), dst.shape[0], None if n_valid is None else n_valid.data_ptr(),
                                                    self._rows2d(src, "src"), self._stream(src.device)), "fl_ep_gather_rows_div")

    def send(self, x, send_slot, per_token, send_buf):
        """send_buf[send_slot[p]] = x[p // per_token] for every entry p with a slot; send_buf [rows, hidden] (view ok)"""
        self._check(self._lib.fl_ep_send_rows(x.data_ptr(), x.shape[0], send_slot.data_ptr(), send_slot.numel(), per_token, x.shape[1],
                                              send_buf.data_ptr(), send_buf.shape[0], self._rows2d(send_buf, "send_buf"),
                                              self._stream(send_buf.device)), "fl_ep_send_rows")

    def combine(self, rows, slot, weights, out, per_token):
        """out[t] = sum_j weights[t, j] * rows[slot[t, j]] (slots < 0 or >= len(rows): skipped), fp32 accumulate; weights: f32
        [tokens, per_token] (view ok) or a flat tensor"""
        w_stride = self._rows2d(weights, "weights") if weights.dim() == 2 else per_token
        self._check(self._lib.fl_ep_combine(rows.data_ptr(), rows.shape[0], slot.data_ptr(), weights.data_ptr(),
                                            out.shape[0], per_token, out.shape[1], out.data_ptr(), w_stride, self._stream(out.device)),
                    "fl_ep_combine")

    def gather_f32(self, vals, src, out):
        """out[r, j] = vals[src[r * cols + j]] where src names a value, else 0; out: f32 [rows, cols] (view ok)"""
        self._check(self._lib.fl_ep_gather_f32(vals.data_ptr(), vals.numel(), src.data_ptr(), out.data_ptr(), out.shape[0] * out.shape[1],
                                               out.shape[1], self._rows2d(out, "out"), self._stream(out.device)), "fl_ep_gather_f32")


class AllToAll:
    """AllToAll(top_k, num_experts, hidden_size, max_tokens, comm_ptr): `max_tokens` is the GLOBAL token capacity
    (low_latency_max_num_tokens_per_gpu * world, fast_ep.py:20); `comm_ptr` (MSCCL++ communicator of the reference) is
    the `data_ptr()` of an eps.communication.MscclppCommunicator — here a host object naming the torch.distributed group the exchange
    runs on (`group` overrides it; default: the world group)."""

    def __init__(self, top_k, num_experts, hidden_size, max_tokens, comm_ptr=None, group=None, row_ops=None):
        self.top_k, self.num_experts, self.hidden = int(top_k), int(num_experts), int(hidden_size)
        from .comm import communicator_from_ptr

        self.communicator = communicator_from_ptr(comm_ptr)
        if group is None and self.communicator is not None:
            group = self.communicator.group
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if self.num_experts % self.world:
            raise RuntimeError("num_experts must divide evenly over the EP group")
        if self.world > 64:
            raise RuntimeError("AllToAll: at most 64 ranks (peer sets are 64-bit masks)")
        self.experts_per_rank = self.num_experts // self.world
        self.max_tokens_per_rank = max(1, int(max_tokens) // self.world)
        self.cap = self.max_tokens_per_rank                       # rows of one peer slab: one row per (token, peer)
        self.row_ops = row_ops if row_ops is not None else HipRowOps()
        self._state = None
        self._ones = None
        # decode-sized exchanges on the peer-mapped one-shot transport (collective, all-or-nothing construction like TPDPConvertor's:
        # every rank builds its AllToAll at the same point of start-up, fast_ep.py:16-22); FLUENT_ONESHOT=0 disables, =1 also builds it
        # at world 1 (tests)
        import os

        self.oneshot = None
        self.messages = {"oneshot": 0, "rccl": 0}                 # launches per route (tests, bench `comm_route`)
        # OPT-IN (FLUENT_EP_ONESHOT=1): the EP exchange on this transport has only ever run as several processes on ONE GPU over hipIpc
        # (profiles/r04_ep_oneshot_two_and_three_processes_one_gpu.txt); until it has passed a real multi-GPU xGMI run the default route is
        # RCCL all_to_all_single — a fault there is an error, not a 10 s spin followed by poisoned rows.  FLUENT_ONESHOT=0
        # still disables every one-shot route; FLUENT_ONESHOT=1 builds the transport at world 1 as well (tests).
        want = os.environ.get("FLUENT_ONESHOT", "auto")
        ep_optin = os.environ.get("FLUENT_EP_ONESHOT", "0") == "1"
        multi = dist.is_initialized() and self.world > 1 and dist.get_backend(group) == "nccl"
        tail = (4 * self.top_k + 7) // 8 * 8
        if row_ops is None and ep_optin and want != "0" and torch.cuda.is_available() and (multi or want == "1") \
                and self.world * self.cap <= 1024 and self.hidden + tail <= 8192 and self.hidden % 8 == 0:
            from .oneshot import OneShotComm
            try:
                self.oneshot = OneShotComm(self.rank if multi else 0, self.world if multi else 1, self.cap, self.hidden + tail, group=group)
            except RuntimeError as ex:
                import warnings

                warnings.warn(f"fluent_mi355: one-shot EP exchange unavailable ({ex}); using RCCL all_to_all_single")

    @property
    def comm_route(self):
        return "one-shot peer-mapped kernel (rows pushed into the peers' inboxes)" if self.oneshot is not None else \
            "RCCL all_to_all_single"

    def slab_bytes(self, dtype_size=2):
        """bytes one rank puts on the wire per direction (rows only)"""
        return self.world * self.cap * self.hidden * dtype_size

    def _a2a(self, inp, ids_col=-1):
        """slab p of `inp` -> slab `rank` of rank p's result.  ids_col (dispatch message): 2-byte-element index of the row's expert ids"""
        if self.oneshot is not None and inp.is_cuda:
            out = torch.empty_like(inp)
            if self.oneshot.accepts_alltoall(inp, out, self.cap):
                self.oneshot.alltoall(inp, out, self.cap, ids_col, self.top_k if ids_col >= 0 else 0)
                self.messages["oneshot"] += 1
                return out
        if self.world == 1:
            return inp                                            # one rank: the slab it sends is the slab it receives
        out = torch.empty_like(inp)
        dist.all_to_all_single(out, inp, group=self.group)       # equal splits: world slabs of `cap` rows
        self.messages["rccl"] += 1
        return out

    def _message(self, dtype, device):
        """one slab-row message buffer [S, hidden + tail] in the row dtype and its views: rows, ids (int32), weights (f32)"""
        K = self.top_k
        tail = (4 * K + 7) // 8 * 8                                # 2-byte elements behind the row: K int32 ids + K f32 weights, 16-B rounded
        S = self.world * self.cap
        msg = torch.empty(S, self.hidden + tail, dtype=dtype, device=device)
        h2 = self.hidden // 2
        return msg, msg[:, :self.hidden], msg.view(torch.int32)[:, h2:h2 + K], msg.view(torch.float32)[:, h2 + K:h2 + 2 * K]

    def dispatch(self, out_exclusive_sum, out_expert_x, dp_x, indices, num_global_tokens, weights=None):
        """`weights` (optional, an extension over fast_ep.py:45-51): the routing weights [tokens, top_k] of the SAME step — they then
        travel in the dispatch message and `combine` needs no message of its own for them (one all-to-all per direction)."""
        t = dp_x.shape[0]
        # (the kernels copy 2-byte rows and read int32 ids whatever the tensors claim to be: check before marshalling)
        if indices.dtype != torch.int32:
            raise RuntimeError(f"AllToAll.dispatch: indices must be int32 (got {indices.dtype})")
        if dp_x.dim() != 2 or dp_x.shape[1] != self.hidden or dp_x.element_size() != 2:
            raise RuntimeError(f"AllToAll.dispatch: dp_x must be a 2-byte [tokens, {self.hidden}] tensor (got {tuple(dp_x.shape)} {dp_x.dtype})")
        if indices.numel() != t * self.top_k:
            raise RuntimeError(f"AllToAll.dispatch: indices must hold {t} x {self.top_k} expert ids (got {indices.numel()})")
        if self.hidden % 8:
            raise RuntimeError("AllToAll: hidden must be a multiple of 8")
        if t > self.max_tokens_per_rank:
            raise RuntimeError(f"{t} local tokens exceed the capacity {self.max_tokens_per_rank} this AllToAll was built for")
        if weights is not None and weights.numel() != t * self.top_k:
            raise RuntimeError("AllToAll.dispatch: weights must hold tokens x top_k values")
        dev, W, K = dp_x.device, self.world, self.top_k
        S = W * self.cap
        idx = indices.reshape(-1).contiguous()
        tok_slot = torch.empty(t * W, dtype=torch.int32, device=dev)
        pair_src = torch.empty(S * K, dtype=torch.int32, device=dev)
        msg, send_rows, send_eid, send_w = self._message(dp_x.dtype, dev)
        if weights is not None:   # ids and weights into the row tails in ONE launch
            self.row_ops.route_dedup(idx, K, self.experts_per_rank, W, self.cap, tok_slot, send_eid, pair_src,
                                     weights.to(torch.float32).reshape(-1).contiguous(), send_w)
        else:
            self.row_ops.route_dedup(idx, K, self.experts_per_rank, W, self.cap, tok_slot, send_eid, pair_src)
        self.row_ops.send(dp_x.contiguous(), tok_slot, W, send_rows)              # empty rows: never read (all their ids are -1)
        recv = self._a2a(msg, ids_col=self.hidden)                                # ONE message: rows + ids (+ weights)
        h2 = self.hidden // 2
        recv_rows, recv_eid = recv[:, :self.hidden], recv.view(torch.int32)[:, h2:h2 + K]
        recv_w = recv.view(torch.float32)[:, h2 + K:h2 + 2 * K] if weights is not None else None
        # received (row, j) pairs grouped by local expert; a row with several local experts is replicated HERE
        order = torch.empty(S * K, dtype=torch.int32, device=dev)
        inv = torch.empty(S * K, dtype=torch.int32, device=dev)      # position of every received pair in the sorted rows (combine)
        self.row_ops.sort(recv_eid, self.experts_per_rank, order, out_exclusive_sum, inv)
        n_out = min(out_expert_x.shape[0], S * K)
        # (static launch over the row bound; only the rows below exclusive_sum[-1] — a device value — are copied)
        self.row_ops.gather_div(recv_rows, order, n_out, K, out_expert_x, out_exclusive_sum[self.experts_per_rank:])
        self._state = (tok_slot, pair_src, inv, n_out, S, recv_w)
        return out_expert_x, out_exclusive_sum

    def combine(self, out_tokens, weights, expert_y, num_global_tokens):
        if self._state is None:
            raise RuntimeError("combine() without a preceding dispatch()")
        tok_slot, pair_src, inv, n_out, S, recv_w = self._state
        dev, W, K = expert_y.device, self.world, self.top_k
        if expert_y.element_size() != 2 or expert_y.shape[1] != self.hidden:
            raise RuntimeError(f"AllToAll.combine: expert_y must be a 2-byte [rows, {self.hidden}] tensor")
        if weights.numel() != out_tokens.shape[0] * K:
            raise RuntimeError("AllToAll.combine: weights must hold tokens x top_k values")
        if recv_w is None:
            # the weights did not travel with the dispatch (the reference's call order): they go to the expert ranks now, in the
            # layout of the expert ids (0 where a row has no j-th expert)
            send_w = torch.empty(S, K, dtype=torch.float32, device=dev)
            self.row_ops.gather_f32(weights.to(torch.float32).reshape(-1).contiguous(), pair_src, send_w)
            recv_w = self._a2a(send_w)
        # expert rank: ONE row per received token = weighted sum over its local experts (rows nobody computed — pairs beyond
        # expert_y — and empty slab rows contribute nothing: their weight is 0 / their position is out of range)
        back = torch.empty(S, self.hidden, dtype=expert_y.dtype, device=dev)
        self.row_ops.combine(expert_y[:n_out] if n_out < expert_y.shape[0] else expert_y, inv, recv_w, back, K)
        ret = self._a2a(back)
        # home rank: sum of the (at most `world`) rows that came back for every token
        if self._ones is None or self._ones.numel() < tok_slot.numel() or self._ones.device != dev:
            self._ones = torch.ones(max(tok_slot.numel(), 1), dtype=torch.float32, device=dev)
        self.row_ops.combine(ret, tok_slot, self._ones, out_tokens, W)
        return out_tokens


# ---- DeepExecutor's ep_scatter / ep_gather (python/sglang/srt/layers/moe/executors/deep_ep_executor.py:271-332,396-430) ----
def _ep_sg_lib():
    import ctypes

    from ._lib import check, lib, stream_ptr

    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    lib.fl_ep_scatter.argtypes = [vp, i64, vp, i64, vp, i32, i64, i64, i32, i32, vp, i32, vp, vp, i64, vp, i64, vp, i64, vp, i64, vp]
    lib.fl_ep_gather.argtypes = [vp, i64, i64, vp, i32, i64, vp, i64, vp, i64, i64, i32, i32, vp, i64, vp]
    lib.fl_ep_scatter.restype = lib.fl_ep_gather.restype = i32
    return check, lib, stream_ptr


def _ids(t, name):
    if t.dtype not in (torch.int32, torch.int64) or t.stride(1) != 1:
        raise RuntimeError(f"{name} must be int32 / int64 with a contiguous last dimension")
    return int(t.dtype == torch.int64)


@torch.no_grad()
def ep_scatter(recv_x, recv_x_scale, recv_topk, num_recv_tokens_per_expert, expert_start_loc, output_tensor,
               output_tensor_scale, m_indices, output_index):
    """Same signature and effects as the reference's ep_scatter (deep_ep_executor.py:271-332): fills expert_start_loc,
    m_indices, output_tensor(+scale) and output_index in place."""
    check, lib, stream_ptr = _ep_sg_lib()
    if recv_x.element_size() != 1 or recv_x.stride(1) != 1 or output_tensor.element_size() != 1 or output_tensor.stride(1) != 1:
        raise RuntimeError("ep_scatter: recv_x / output_tensor must be fp8 rows with a contiguous last dimension")
    if recv_x_scale.dtype != torch.float32 or output_tensor_scale.dtype != torch.float32 or recv_x_scale.stride(1) != 1 \
            or output_tensor_scale.stride(1) != 1:
        raise RuntimeError("ep_scatter: scales must be float32 with a contiguous last dimension")
    if m_indices.shape[0] % 128:
        raise RuntimeError("ep_scatter: m_indices must be a multiple of 128 rows (deep_ep_executor.py:290)")
    for t, n in ((num_recv_tokens_per_expert, "num_recv_tokens_per_expert"), (expert_start_loc, "expert_start_loc"),
                 (m_indices, "m_indices"), (output_index, "output_index")):
        if t.dtype != torch.int32:
            raise RuntimeError(f"ep_scatter: {n} must be int32")
    T, K = recv_topk.shape
    check(lib.fl_ep_scatter(recv_x.data_ptr(), recv_x.stride(0), recv_x_scale.data_ptr(), recv_x_scale.stride(0),
                            recv_topk.data_ptr(), _ids(recv_topk, "recv_topk"), recv_topk.stride(0), T, K, recv_x.shape[1],
                            num_recv_tokens_per_expert.data_ptr(), num_recv_tokens_per_expert.shape[0],
                            expert_start_loc.data_ptr(), output_tensor.data_ptr(), output_tensor.stride(0),
                            output_tensor_scale.data_ptr(), output_tensor_scale.stride(0), m_indices.data_ptr(),
                            min(output_tensor.shape[0], m_indices.shape[0]), output_index.data_ptr(), output_index.stride(0),
                            HipRowOps._stream_of(recv_x)), "fl_ep_scatter")


@torch.no_grad()
def ep_gather(input_tensor, recv_topk_ids, recv_topk_weight, input_index, output_tensor):
    """Same signature and effect as the reference's ep_gather (deep_ep_executor.py:396-430): output_tensor[t] = sum over the
    local experts of token t of weight * input_tensor[input_index[t, k]] (fp32 accumulate, bf16 out)."""
    check, lib, stream_ptr = _ep_sg_lib()
    if input_tensor.dtype != torch.bfloat16 or output_tensor.dtype != torch.bfloat16 or input_tensor.stride(1) != 1 \
            or output_tensor.stride(1) != 1:
        raise RuntimeError("ep_gather: input / output must be bf16 with a contiguous last dimension")
    if recv_topk_weight.dtype != torch.float32 or input_index.dtype != torch.int32:
        raise RuntimeError("ep_gather: weights must be float32 and input_index int32")
    T, K = recv_topk_ids.shape
    check(lib.fl_ep_gather(input_tensor.data_ptr(), input_tensor.stride(0), input_tensor.shape[0], recv_topk_ids.data_ptr(),
                           _ids(recv_topk_ids, "recv_topk_ids"), recv_topk_ids.stride(0), recv_topk_weight.data_ptr(),
                           recv_topk_weight.stride(0), input_index.data_ptr(), input_index.stride(0), output_tensor.shape[0], K,
                           input_tensor.shape[1], output_tensor.data_ptr(), output_tensor.stride(0),
                           HipRowOps._stream_of(input_tensor)), "fl_ep_gather")
