"""GPU box: the one-shot peer-mapped exchange between TWO PROCESSES on ONE GPU (the boxes here have a single GPU: this is
the closest hardware run of the multi-rank kernel path — real hipIpc mapping, real cross-process flag waits; what it
cannot show is xGMI).  Handles travel over a gloo group.  Every wait is time-bounded, so a scheduling problem shows up as
an error, not a hang.  usage: python tools/oneshot_two_procs.py [world]"""
import os, sys, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch, torch.distributed as dist, torch.multiprocessing as mp


def worker(rank, world, port, T, H, iters):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from fluent_mi355.oneshot import OneShotComm
    from fluent_mi355.comm import HipNormOps, get_num_tokens_per_rank
    c = OneShotComm(rank, world, 64, H, timeout_s=5.0)
    ops = HipNormOps()
    ok = True
    for it in range(iters):
        g = torch.Generator().manual_seed(100 * it)
        xs = [torch.randn(T, H, generator=g).to(torch.bfloat16) for _ in range(world)]      # every rank knows all inputs
        res = torch.randn(T, H, generator=g).to(torch.bfloat16).to(dev)
        gamma = torch.rand(H, generator=g).to(torch.bfloat16).to(dev)
        pieces = torch.stack(xs).to(dev)
        # expected: the RCCL route's kernel on the stacked pieces
        e_res, e_norm = torch.empty(T, H, dtype=torch.bfloat16, device=dev), torch.empty(T, H, dtype=torch.bfloat16, device=dev)
        ops.add_rmsnorm(pieces, None, res, gamma, 1e-6, e_res, e_norm, None, None)
        o_res, o_norm = torch.empty_like(e_res), torch.empty_like(e_norm)
        c.allreduce_fused(xs[rank].to(dev), res, gamma, 1e-6, o_res, o_norm)
        counts = get_num_tokens_per_rank(world, T)
        lo = sum(counts[:rank]); hi = lo + counts[rank]
        r_res, r_norm = torch.empty(hi - lo, H, dtype=torch.bfloat16, device=dev), torch.empty(hi - lo, H, dtype=torch.bfloat16, device=dev)
        c.reducescatter_fused(xs[rank].to(dev), None, res[lo:hi].contiguous(), gamma, 1e-6, r_res, r_norm)
        # one-shot all-gather (C3) and all-gather + dual RMSNorm (C7) of the uneven row split, against the kernel of the RCCL route
        D, QR, KVR = 2112, 1536, 512
        full = torch.randn(T, D, generator=g).to(torch.bfloat16).to(dev)
        gq, gkv = torch.rand(QR, generator=g).to(torch.bfloat16).to(dev), torch.rand(KVR, generator=g).to(torch.bfloat16).to(dev)
        ag = torch.zeros(T, D, dtype=torch.bfloat16, device=dev)
        c.allgather_fused(full[lo:hi].contiguous(), T, ag)
        e_ag, e_x = full.clone(), torch.empty(T, QR, dtype=torch.bfloat16, device=dev)
        e_q, e_s = torch.empty(T, QR, dtype=torch.float8_e4m3fn, device=dev), torch.empty(T, QR // 128, device=dev)
        ops.dual_rmsnorm(e_ag, QR, KVR, gq, gkv, 1e-6, 1e-6, e_x, e_q, e_s)
        ag2, x2 = torch.zeros_like(ag), torch.zeros_like(e_x)
        q2, s2 = torch.zeros_like(e_q), torch.zeros_like(e_s)
        c.allgather_fused(full[lo:hi].contiguous(), T, ag2, QR, KVR, gq, gkv, 1e-6, 1e-6, x2, q2, s2)
        c.check()
        ok &= torch.equal(o_res, e_res) and torch.equal(o_norm, e_norm) and torch.equal(r_res, e_res[lo:hi]) and torch.equal(r_norm, e_norm[lo:hi])
        ok &= torch.equal(ag, full) and torch.equal(ag2, e_ag) and torch.equal(x2, e_x)
        ok &= torch.equal(q2.view(torch.uint8), e_q.view(torch.uint8)) and torch.equal(s2, e_s)
    # latency of the fused all-reduce (graph of 20 launches)
    x = xs[rank].to(dev)
    for _ in range(3): c.allreduce_fused(x, res, gamma, 1e-6, o_res, o_norm)
    torch.cuda.synchronize(); dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): c.allreduce_fused(x, res, gamma, 1e-6, o_res, o_norm)
    e1.record(); torch.cuda.synchronize()
    c.check()
    print(f"rank {rank}/{world}: T={T} H={H} parity {'OK' if ok else 'MISMATCH'} over {iters} iterations; fused all-reduce {e0.elapsed_time(e1) / 50 * 1e3:.1f} us/op (processes sharing one GPU)", flush=True)
    dist.barrier()
    c.close()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


def ep_worker(rank, world, port, tokens, hidden, iters):
    """C1 / C2 on the one-shot transport: eps.fast_ep.AllToAll.dispatch -> per-expert op -> combine between `world` processes on one
    GPU, every exchange ONE oneshot_a2a_kernel launch over hipIpc-mapped inboxes, against the SAME host logic and row kernels with the
    exchange staged through a gloo all_to_all_single on the CPU: bit-identical outputs (only the transport differs)."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from fluent_mi355.ep import AllToAll
    from fluent_mi355.oneshot import OneShotComm
    K, E = 8, 32 * world
    el = E // world

    def gloo_a2a(self, inp, ids_col=-1):
        src = inp.cpu().contiguous()
        out = torch.empty_like(src)
        dist.all_to_all_single(out, src)
        self.messages["rccl"] += 1
        return out.to(inp.device)

    a_one = AllToAll(K, E, hidden, tokens * world, None)
    a_one.oneshot = OneShotComm(rank, world, a_one.cap, hidden + (4 * K + 7) // 8 * 8, timeout_s=10.0)
    a_ref = AllToAll(K, E, hidden, tokens * world, None)
    a_ref._a2a = gloo_a2a.__get__(a_ref)
    ok = True
    rows = world * a_one.cap * K
    for it in range(iters):
        g = torch.Generator().manual_seed(1000 * it + rank)
        t = tokens if it % 3 else max(1, tokens // 3)             # fewer tokens than the capacity: mostly empty slabs
        x = torch.randn(t, hidden, generator=g).to(torch.bfloat16).to(dev)
        ids = torch.stack([torch.randperm(E, generator=g)[:K] for _ in range(t)]).to(torch.int32)
        if it == 1:
            ids = (torch.arange(K, dtype=torch.int32) + el * ((rank + 1) % world)).repeat(t, 1)    # every token to ONE peer: full slab
        ids = ids.to(dev)
        w = torch.rand(t, K, generator=g).to(dev)
        outs = []
        for a, with_w in ((a_one, True), (a_ref, True), (a_one, False), (a_ref, False)):
            ex = torch.zeros(el + 1, dtype=torch.int32, device=dev)
            xr = torch.zeros(rows, hidden, dtype=torch.bfloat16, device=dev)
            a.dispatch(ex, xr, x, ids, t * world, weights=w if with_w else None)
            y = (xr.float() * 1.5 + 0.25).to(torch.bfloat16)        # stands in for the expert MLPs (row-wise, routing-independent)
            out = torch.zeros(t, hidden, dtype=torch.bfloat16, device=dev)
            a.combine(out, w, y, t * world)
            torch.cuda.synchronize()
            outs.append((out.clone(), ex.clone()))
        a_one.oneshot.check()
        ok &= torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        ok &= torch.equal(outs[2][0], outs[3][0]) and torch.equal(outs[0][0], outs[2][0])
        # and against plain arithmetic: out[t] = sum_k w[t,k] * (1.5 x[t] + 0.25), rounded once per (token, rank) + once at home
        want = (w.sum(1, keepdim=True) * (x.float() * 1.5 + 0.25).to(torch.bfloat16).float())
        ok &= bool(((outs[0][0].float() - want).abs() <= 0.03 * want.abs() + 0.05).all())
    ok &= a_one.messages["rccl"] == 0 and a_one.messages["oneshot"] == iters * (2 + 3)   # with weights: 2 launches per layer; without: 3
    # latency of one dispatch-sized exchange (msg rows + tail) on the transport alone
    msg = torch.zeros(world * a_one.cap, hidden + 32, dtype=torch.bfloat16, device=dev)
    msg.view(torch.int32)[:, hidden // 2:hidden // 2 + K] = 1
    out = torch.empty_like(msg)
    for _ in range(3): a_one.oneshot.alltoall(msg, out, a_one.cap, hidden, K)
    torch.cuda.synchronize(); dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): a_one.oneshot.alltoall(msg, out, a_one.cap, hidden, K)
    e1.record(); torch.cuda.synchronize()
    a_one.oneshot.check()
    print(f"rank {rank}/{world}: EP dispatch/combine on the one-shot transport {'OK' if ok else 'MISMATCH'} over {iters} iterations "
          f"({a_one.messages}); all-to-all of {world} x {a_one.cap} full rows x {hidden + 32} bf16: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us/op "
          f"(processes sharing one GPU)", flush=True)
    dist.barrier()
    a_one.oneshot.close()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


def lost_peer_worker(rank, world, port, T, H):
    """rank 1 skips one operation: rank 0's wait runs out of its (here 1 s) budget.  Required behaviour: the
    failing launch leaves NaN in every output row, never partial sums; the epoch does not advance; the next launch call
    raises; a rank whose peer is dead fails the same way one operation later — nothing hangs, nothing is silently wrong."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from fluent_mi355.oneshot import OneShotComm
    c = OneShotComm(rank, world, 64, H, timeout_s=1.0)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(T, H, generator=g).to(torch.bfloat16).to(dev)
    res = torch.randn(T, H, generator=g).to(torch.bfloat16).to(dev)
    gamma = torch.rand(H, generator=g).to(torch.bfloat16).to(dev)
    o_res, o_norm = torch.zeros(T, H, dtype=torch.bfloat16, device=dev), torch.zeros(T, H, dtype=torch.bfloat16, device=dev)
    c.allreduce_fused(x, res, gamma, 1e-6, o_res, o_norm)          # both ranks: fine
    torch.cuda.synchronize(); dist.barrier()
    ok = bool(torch.isfinite(o_norm.float()).all())
    if rank == 0:
        o_res.zero_(); o_norm.zero_()
        c.allreduce_fused(x, res, gamma, 1e-6, o_res, o_norm)      # the peer never issues this one
        torch.cuda.synchronize()
        ok &= bool(torch.isnan(o_res.float()).all()) and bool(torch.isnan(o_norm.float()).all())
        raised = False
        try:
            c.allreduce_fused(x, res, gamma, 1e-6, o_res, o_norm)  # dead communicator: the launch call itself reports it
        except RuntimeError as ex:
            raised = "timed out" in str(ex)
        ok &= raised
        try:
            c.check(); ok = False
        except RuntimeError:
            pass
    dist.barrier()
    if rank != 0:
        # the late rank: operation 2 still completes — rank 0 pushed its (valid) rows and flags for it before it gave up —
        # but operation 3 finds nobody (rank 0 is dead at epoch 2 and pushes nothing any more): NaN, not a hang
        c.allreduce_fused(x, res, gamma, 1e-6, o_res, o_norm)
        torch.cuda.synchronize()
        ok &= bool(torch.isfinite(o_norm.float()).all())
        o_res.zero_(); o_norm.zero_()
        c.allreduce_fused(x, res, gamma, 1e-6, o_res, o_norm)
        torch.cuda.synchronize()
        ok &= bool(torch.isnan(o_norm.float()).all())
    print(f"rank {rank}/{world}: lost-peer handling {'OK' if ok else 'WRONG'}", flush=True)
    dist.barrier()
    c.close()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


def beside_k1_worker(rank, world, port, T, H):
    """C6 / C5 on one stream beside K1 on another (LongCat's two-stream layer, models/longcat_flash.py:417-445).
    K1 fills every CU's registers and LDS (8 waves x 256 VGPRs, 160 KB), so the one-shot kernel's push workgroups queue
    behind K1 workgroups while the peer spins on their flags: that must cost latency, never a timeout or a wrong sum.  Each
    rank keeps a stream busy with cfg2-shaped K1 launches and runs fused all-reduces / reduce-scatters on the main stream."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    sys.path.insert(0, ROOT)
    import bench
    import flash_mla_fp8 as fm
    from fluent_mi355.oneshot import OneShotComm
    from fluent_mi355.comm import HipNormOps
    c = OneShotComm(rank, world, 64, H, timeout_s=30.0)
    ops = HipNormOps()
    wl = bench.build_workload(dev, 2, 64, 2048, 128, seed=3 + rank)
    meta, ns = fm.get_mla_metadata(wl["seqlens"], 128, 1)
    side = torch.cuda.Stream()
    stop_after = 150
    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(T, H, generator=g).to(torch.bfloat16) for _ in range(world)]
    res = torch.randn(T, H, generator=g).to(torch.bfloat16).to(dev)
    gamma = torch.rand(H, generator=g).to(torch.bfloat16).to(dev)
    e_res, e_norm = torch.empty(T, H, dtype=torch.bfloat16, device=dev), torch.empty(T, H, dtype=torch.bfloat16, device=dev)
    ops.add_rmsnorm(torch.stack(xs).to(dev), None, res, gamma, 1e-6, e_res, e_norm, None, None)
    x = xs[rank].to(dev)
    o_res, o_norm = torch.empty_like(e_res), torch.empty_like(e_norm)
    torch.cuda.synchronize(); dist.barrier()
    ok = True
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        for i in range(stop_after):
            bench.layer_call(fm, wl, i & 1, meta, ns)
    e0.record()
    for i in range(60):
        c.allreduce_fused(x, res, gamma, 1e-6, o_res, o_norm)
        if i % 10 == 9:
            torch.cuda.current_stream().synchronize()
            ok &= torch.equal(o_res, e_res) and torch.equal(o_norm, e_norm)
    e1.record()
    torch.cuda.synchronize()
    c.check()
    print(f"rank {rank}/{world}: beside K1 {'OK' if ok else 'MISMATCH'}; fused all-reduce {e0.elapsed_time(e1) / 60 * 1e3:.1f} us/op while a second "
          f"stream runs bs=64 seq=2048 H=128 MLA decode launches (processes sharing one GPU)", flush=True)
    dist.barrier()
    c.close()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    if len(sys.argv) > 2 and sys.argv[2] == "lostpeer":
        mp.spawn(lost_peer_worker, args=(2, port, 16, 2048), nprocs=2, join=True)
    elif len(sys.argv) > 2 and sys.argv[2] == "ep":
        mp.spawn(ep_worker, args=(world, port, 32, 7168, 6), nprocs=world, join=True)
    elif len(sys.argv) > 2 and sys.argv[2] == "k1":
        mp.spawn(beside_k1_worker, args=(2, port, 48, 7168), nprocs=2, join=True)
    else:
        mp.spawn(worker, args=(world, port, 49 if world == 2 else 48, 7168, 5), nprocs=world, join=True)   # 49 rows over 2 ranks: 25 / 24
