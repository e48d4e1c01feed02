"""CPU, world_size 2, gloo: the N>1 host logic of the path — EP all-to-all dispatch/combine (fast_ep.py:45-51,73-78
semantics) and the bench's max-over-ranks timing reduction."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, t_per_rank, one_peer=False, weights_at_dispatch=False):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "sglang-fluentllm_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("FLUENT_MI355_ALLOW_MISSING_LIB", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ep_torch_ops import TorchRowOps
        from fluent_mi355.ep import AllToAll

        E, K, HID = 8, 3, 64
        g = torch.Generator().manual_seed(100 + rank)
        t = t_per_rank[rank]
        T_g = sum(t_per_rank)
        x = torch.randn(t, HID, generator=g).to(torch.bfloat16)
        idx = torch.stack([torch.randperm(E, generator=g)[:K] for _ in range(t)]).to(torch.int32) if t else torch.zeros(0, K, dtype=torch.int32)
        if one_peer and t:                                     # worst case of the slab sizing: EVERY token of every rank goes to rank 1
            idx = torch.stack([torch.randperm(E // world, generator=g)[:K] + E // world for _ in range(t)]).to(torch.int32)
        w = torch.rand(t, K, generator=g)
        a2a = AllToAll(K, E, HID, max(t_per_rank) * world, None, row_ops=TorchRowOps())
        assert a2a.cap == max(t_per_rank) and a2a.slab_bytes() == world * max(t_per_rank) * HID * 2   # one row per (token, peer)
        ex = torch.empty(E // world + 1, dtype=torch.int32)
        expert_x = torch.zeros(T_g * K, HID, dtype=torch.bfloat16)
        sent = []
        orig_a2a = a2a._a2a
        a2a._a2a = lambda inp, ids_col=-1: (sent.append(tuple(inp.shape)), orig_a2a(inp, ids_col))[1]      # count the messages of each direction
        a2a.dispatch(out_exclusive_sum=ex, out_expert_x=expert_x, dp_x=x, indices=idx, num_global_tokens=T_g,
                     **({"weights": w} if weights_at_dispatch else {}))
        assert len(sent) == 1 and sent[0][1] == HID + 16, sent                          # ONE dispatch message: row + 3 ids + 3 weights, 16-B rounded
        # exclusive_sum consistent with what every rank routed to my experts
        all_idx = [None] * world
        dist.all_gather_object(all_idx, idx.tolist())
        flat = [e for r in all_idx for row in r for e in row]
        mine = [sum(1 for e in flat if e == rank * (E // world) + le) for le in range(E // world)]
        assert ex.tolist() == [0] + torch.cumsum(torch.tensor(mine), 0).tolist(), (ex.tolist(), mine)
        # "expert compute": local expert le scales its rows by (global expert id + 1)
        y = torch.full_like(expert_x, float("nan"))            # rows no expert computed must never reach a token
        for le in range(E // world):
            lo, hi = int(ex[le]), int(ex[le + 1])
            y[lo:hi] = (expert_x[lo:hi].float() * (rank * (E // world) + le + 1)).to(torch.bfloat16)
        out = torch.empty(t, HID, dtype=torch.bfloat16)
        a2a.combine(out_tokens=out, weights=w, expert_y=y, num_global_tokens=T_g)
        assert len(sent) == (2 if weights_at_dispatch else 3), sent                     # combine: rows back (+ the weights if they did not travel yet)
        ref = sum(w[:, k:k + 1] * (x.float() * (idx[:, k:k + 1].float() + 1)).to(torch.bfloat16).float() for k in range(K)).to(torch.bfloat16) if t else out
        assert torch.allclose(out.float(), ref.float(), atol=2e-2, rtol=2e-2), float((out.float() - ref.float()).abs().max())
        # bench.py's reduction: step time = MAX over ranks
        tm = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        assert float(tm) == float(world)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("t_per_rank", [[5, 7], [0, 4]])
def test_ep_all_to_all_world2_gloo(t_per_rank):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, t_per_rank), nprocs=2, join=True)


def test_ep_all_to_all_world2_gloo_weights_travel_with_the_dispatch():
    """dispatch(..., weights=): ids AND routing weights in the slab-row tail -> one all-to-all per direction"""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, [5, 7], False, True), nprocs=2, join=True)


def test_ep_all_to_all_world2_gloo_every_token_to_one_peer():
    """The peer slab holds max_tokens_per_rank rows (a token travels to a rank once): full slabs, nothing dropped."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, [6, 6], True), nprocs=2, join=True)


def _replay_worker(rank, world, port):
    """hipGraph-style use: buffers allocated ONCE, three "replays" with different routing written into the same input
    tensors; every intermediate the exchange allocates must have a routing-independent shape (a captured graph replays
    fixed launches on fixed shapes), and each replay must give that replay's answer."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "sglang-fluentllm_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ep_torch_ops import TorchRowOps
        from fluent_mi355.ep import AllToAll

        E, K, HID, t = 8, 3, 64, 6
        a2a = AllToAll(K, E, HID, t * world, None, row_ops=TorchRowOps())
        x = torch.empty(t, HID, dtype=torch.bfloat16)
        idx = torch.empty(t, K, dtype=torch.int32)
        w = torch.empty(t, K)
        ex = torch.empty(E // world + 1, dtype=torch.int32)
        expert_x = torch.zeros(world * a2a.cap * K, HID, dtype=torch.bfloat16)      # the static row bound
        out = torch.empty(t, HID, dtype=torch.bfloat16)
        shapes = None
        for it in range(3):
            g = torch.Generator().manual_seed(1000 * it + rank)
            x.copy_(torch.randn(t, HID, generator=g).to(torch.bfloat16))
            if it == 1:      # skewed: everything to rank 0's experts
                idx.copy_(torch.stack([torch.randperm(E // world, generator=g)[:K] for _ in range(t)]).to(torch.int32))
            else:
                idx.copy_(torch.stack([torch.randperm(E, generator=g)[:K] for _ in range(t)]).to(torch.int32))
            w.copy_(torch.rand(t, K, generator=g))
            a2a.dispatch(out_exclusive_sum=ex, out_expert_x=expert_x, dp_x=x, indices=idx, num_global_tokens=t * world)
            now = [tuple(v.shape) if torch.is_tensor(v) else v for v in a2a._state]
            assert shapes is None or shapes == now, (shapes, now)
            shapes = now
            y = torch.full_like(expert_x, float("nan"))
            for le in range(E // world):
                lo, hi = int(ex[le]), int(ex[le + 1])
                y[lo:hi] = (expert_x[lo:hi].float() * (rank * (E // world) + le + 1)).to(torch.bfloat16)
            a2a.combine(out_tokens=out, weights=w, expert_y=y, num_global_tokens=t * world)
            ref = sum(w[:, k:k + 1] * (x.float() * (idx[:, k:k + 1].float() + 1)).to(torch.bfloat16).float() for k in range(K))
            assert torch.allclose(out.float(), ref.to(torch.bfloat16).float(), atol=2e-2, rtol=2e-2), it
    finally:
        dist.destroy_process_group()


def test_ep_all_to_all_static_shape_replays_world2_gloo():
    port = _free_port()
    mp.spawn(_replay_worker, args=(2, port), nprocs=2, join=True)
