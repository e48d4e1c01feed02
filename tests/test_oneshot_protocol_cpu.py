"""CPU, no torch.distributed: the flag / epoch / parity protocol of the one-shot peer-mapped exchange (csrc/comm_protocol.h),
driven through the library's host emulation `fl_comm_host_exchange` by several PROCESSES that share their workspaces
through POSIX shared memory — the layout arithmetic is the code the device kernel compiles (same header), the
push / flag / wait sequence is the kernel's, restated for host memory.  What is checked: sums exact on every rank for
hundreds of back-to-back operations of mixed kind (all-reduce / reduce-scatter) and size — including reduce-scatters
with fewer tokens than ranks, where a rank receives nothing and must still not run ahead into a buffer a slower peer is
reading (the sync row) — under random per-rank delays; and that a missing peer is reported as a timeout, not a hang."""
import ctypes
import multiprocessing as mp
import os
import random
import sys
import time
from multiprocessing import shared_memory

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "sglang-fluentllm_amd"))

MAX_T, HID = 8, 64


def _lib():
    from fluent_mi355._lib import lib
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    lib.fl_comm_workspace_size.argtypes = [i32, i64, i32, ctypes.POINTER(i64)]
    lib.fl_comm_host_init.argtypes = [vp, i32, i64, i32]
    lib.fl_comm_host_exchange.argtypes = [ctypes.POINTER(vp), i32, i32, i64, i32, i32, vp, i64, i32, vp, ctypes.c_double]
    for n in ("fl_comm_workspace_size", "fl_comm_host_init", "fl_comm_host_exchange"):
        getattr(lib, n).restype = i32
    return lib


def _bf16(x):       # float32 -> bf16 bits (values are small integers: exact)
    return (np.asarray(x, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def _plan(world, seed, n_ops):
    """the operation sequence every rank issues: (reduce_scatter, T, H)"""
    r = random.Random(seed)
    return [(r.random() < 0.5, r.choice([0, 1, 2, 3, 5, MAX_T]), r.choice([8, 32, HID])) for _ in range(n_ops)]


def _input(rank, op, T, H):
    return (np.arange(T * H, dtype=np.float32).reshape(T, H) % 7) + rank * 3 + op % 5


def _slice(T, world, r):
    base, rem = divmod(T, world)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)


def _rank_main(rank, world, names, seed, n_ops, jitter, q):
    try:
        lib = _lib()
        shms = [shared_memory.SharedMemory(name=n) for n in names]
        ptrs = (ctypes.c_void_p * world)(*[ctypes.addressof(ctypes.c_char.from_buffer(s.buf)) for s in shms])
        rnd = random.Random(seed * 131 + rank)
        for op, (rs, T, H) in enumerate(_plan(world, seed, n_ops)):
            if jitter and rnd.random() < 0.3:
                time.sleep(rnd.random() * jitter)
            x = _bf16(_input(rank, op, T, H))
            lo, hi = _slice(T, world, rank) if rs else (0, T)
            out = np.full(((hi - lo), H), -1.0, dtype=np.float32)
            st = lib.fl_comm_host_exchange(ptrs, rank, world, MAX_T, HID, int(rs), x.ctypes.data, T, H, out.ctypes.data, 20.0)
            assert st == 0, (op, lib.fl_last_error().decode())
            ref = sum(_input(r, op, T, H) for r in range(world))[lo:hi]
            assert np.array_equal(out, ref), (rank, op, rs, T, H)
        del ptrs
        for s in shms:
            s.close()
        q.put((rank, "ok"))
    except BaseException as ex:   # noqa: BLE001 — reported to the parent
        q.put((rank, f"{type(ex).__name__}: {ex}"))


def _workspaces(lib, world):
    size = ctypes.c_int64()
    assert lib.fl_comm_workspace_size(world, MAX_T, HID, ctypes.byref(size)) == 0
    shms = [shared_memory.SharedMemory(create=True, size=size.value) for _ in range(world)]
    for s in shms:
        assert lib.fl_comm_host_init(ctypes.addressof(ctypes.c_char.from_buffer(s.buf)), world, MAX_T, HID) == 0
    return shms


@pytest.mark.parametrize("world,jitter", [(2, 0.0), (2, 0.004), (3, 0.002)])
def test_oneshot_protocol_processes_over_shared_memory(world, jitter):
    lib = _lib()
    shms = _workspaces(lib, world)
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_rank_main, args=(r, world, [s.name for s in shms], 7, 300, jitter, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = dict(q.get(timeout=120) for _ in range(world))
        for p in procs:
            p.join(timeout=30)
        assert all(v == "ok" for v in res.values()), res
    finally:
        for s in shms:
            s.close()
            s.unlink()


def test_oneshot_protocol_reports_a_missing_peer_as_timeout():
    lib = _lib()
    shms = _workspaces(lib, 2)
    try:
        ptrs = (ctypes.c_void_p * 2)(*[ctypes.addressof(ctypes.c_char.from_buffer(s.buf)) for s in shms])
        x = _bf16(np.ones((2, 8), dtype=np.float32))
        out = np.zeros((2, 8), dtype=np.float32)
        t0 = time.time()
        st = lib.fl_comm_host_exchange(ptrs, 0, 2, MAX_T, HID, 0, x.ctypes.data, 2, 8, out.ctypes.data, 0.2)   # rank 1 never shows up
        assert st != 0 and "timed out" in lib.fl_last_error().decode() and time.time() - t0 < 5
        del ptrs
    finally:
        for s in shms:
            s.close()
            s.unlink()


def test_slice_helper_of_this_file_matches_get_num_tokens_per_rank():
    """the expected slices above (`_slice`) are the reference's token split (flashinfer_comm_fusion.py:237-244); the C side
    (fl_comm_owner / fl_comm_slice_lo) is checked against them by every reduce-scatter of the protocol test"""
    from fluent_mi355.comm import get_num_tokens_per_rank
    for world in (1, 2, 3, 5, 8):
        for T in (0, 1, 2, 7, 8, 9, 64, 130):
            los = [_slice(T, world, r) for r in range(world)]
            assert [hi - lo for lo, hi in los] == get_num_tokens_per_rank(world, T) and los[-1][1] == T
