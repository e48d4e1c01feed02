"""CPU: properties of the split-KV tile scheduler statement (oracle/mla_ref.py:get_mla_metadata = the Python
statement of csrc/mla_metadata.hip; the GPU test compares the two bit-exactly)."""
import random

from oracle import mla_ref


def _coverage(seqlens, parts):
    meta, ns = mla_ref.get_mla_metadata(seqlens, parts)
    bs = len(seqlens)
    nt = [((L + 63) // 64) if L > 0 else 0 for L in seqlens]
    seen = [[0] * n for n in nt]
    touched = [0] * bs
    for p in range(parts):
        b0, t0, b1, t1, s0 = (int(x) for x in meta[p, :5])
        req, tile, split = b0, t0, s0
        while req < bs and (req < b1 or (req == b1 and t1 > 0)):
            te = nt[req] if req < b1 else min(t1, nt[req])
            for t in range(tile, te):
                seen[req][t] += 1
            assert split == touched[req], "split index must count the parts that touched the request before"
            touched[req] += 1
            req, tile, split = req + 1, 0, 0
    assert all(all(c == 1 for c in row) for row in seen), "every tile exactly once"
    for b in range(bs):
        assert ns[b + 1] - ns[b] == max(touched[b], 1) == touched[b], (b, touched[b], ns)
    assert ns[bs] <= bs + parts
    return meta, ns


def test_uniform_decode_batch_has_no_splits():
    meta, ns = _coverage([4096] * 128, 128)
    assert all(ns[b + 1] - ns[b] == 1 for b in range(128))


def test_random_ragged():
    rnd = random.Random(0)
    for _ in range(200):
        bs = rnd.randint(1, 40)
        lens = [rnd.choice([0, 1, 63, 64, 65, rnd.randint(1, 9000)]) for _ in range(bs)]
        _coverage(lens, rnd.choice([1, 2, 7, 64, 128, 256]))


def test_small_batch_splits_long_sequence():
    meta, ns = _coverage([16384], 128)
    assert ns[1] == 32  # one long request is spread over many CUs: max(32, pages/8) parts (256 pages -> 32 x 8 here)
    meta, ns = _coverage([65536], 256)
    assert ns[1] == 128


def test_seed_sweep_covers_the_cases():
    """tools/seed_sweep_mla.py (the source of the K1 tolerance) sweeps exactly the shapes of test_mla_gpu.CASES"""
    import ast
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def literal(path, name):
        tree = ast.parse(open(path).read())
        for node in tree.body:
            if isinstance(node, ast.Assign) and any(getattr(t, "id", None) == name for t in node.targets):
                return ast.literal_eval(node.value)
        raise AssertionError(name)

    assert literal(os.path.join(root, "tools", "seed_sweep_mla.py"), "SHAPES") == literal(
        os.path.join(root, "tests", "test_mla_gpu.py"), "CASES")
