"""CPU: the drop-in boundary's MODULE surface.  tests/golden/import_surface.json (oracle/gen_import_surface.py: an `ast` walk over the
reference's python/sglang) lists every (module, name) the reference takes from the five packages this repo replaces.  Every pair that a
hot-path file (SURVEY section 8 a/b) imports must resolve in the shim under sglang-fluentllm_amd/ — or be listed in INTEGRATION.md's table of
deliberately absent names.  No compute: importing the shim only needs the C-ABI library to load."""
import importlib
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "sglang-fluentllm_amd")


def surface():
    with open(os.path.join(ROOT, "tests", "golden", "import_surface.json")) as f:
        return json.load(f)


def absent_by_design():
    """rows `| `module` | `name` | file:line | why |` of INTEGRATION.md's deliberately-absent table (name `—` = the module itself)"""
    out = set()
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    start = text.index("deliberately absent")
    for m in re.finditer(r"^\s*\| `([\w.]+)` \| (?:`([\w.]+)`|—) \| `srt/", text[start:], re.M):
        out.add((m.group(1), m.group(2)))
    return out


def resolve(module, name):
    parts = module.split(".")
    obj, rest = None, []
    for n in range(len(parts), 0, -1):      # the "module" of an attribute chain may end in a class (eps.communication.MscclppCommunicator)
        try:
            obj = importlib.import_module(".".join(parts[:n]))
            rest = parts[n:]
            break
        except ImportError:
            continue
    if obj is None:
        raise ImportError(module)
    for a in rest + (name.split(".") if name else []):
        obj = getattr(obj, a)
    return obj


@pytest.fixture(scope="module")
def shim_first():
    sys.path.insert(0, SHIM)
    yield
    sys.path.remove(SHIM)


def test_every_hot_path_import_resolves_in_the_shim_or_is_listed_absent(shim_first):
    d = surface()
    assert tuple(d["packages"]) == ("flash_mla_fp8", "flash_mla_swap", "deep_gemm", "flashinfer", "eps") and not d["hot_files_missing"]
    absent = absent_by_design()
    assert len(absent) >= 5
    missing, stale = [], set(absent)
    hot = [r for r in d["records"] if r["hot"]]
    assert len(hot) >= 70
    for r in hot:
        key = (r["module"], r["name"])
        try:
            resolve(*key)
        except (ImportError, AttributeError) as ex:
            if key in absent:
                stale.discard(key)
            else:
                missing.append(f"{r['module']}:{r['name']} ({r['file']}:{r['line']}, {r['level']}): {type(ex).__name__}: {ex}")
        else:
            assert key not in absent, f"{key} resolves in the shim but INTEGRATION.md lists it as deliberately absent"
    assert not missing, "names the reference's hot-path files import that the shim lacks:\n  " + "\n  ".join(sorted(set(missing)))
    assert not stale, f"INTEGRATION.md lists absent names that no hot-path file imports: {sorted(stale)}"


def test_module_level_imports_of_the_hot_path_files_are_all_served(shim_first):
    """The imports that run when the file itself is imported (incl. those under try / if at module level) and are NOT behind a CUDA-only
    probe must all resolve: these are the ones that stop `import sglang.srt.layers.moe.topk` before any kernel is reached."""
    unguarded = {("flashinfer", "topk_softmax"), ("flashinfer", "moe_fused_gate"), ("eps", "utils"), ("eps.utils", "ops._ops.topk_sigmoid"),
                 ("eps.communication", "MscclppCommunicator"), ("eps.communication", "MscclppCommunicatorParams"),
                 ("eps.communication.MscclppCommunicator", "createUniqueId"), ("flash_mla_fp8", None), ("flash_mla_swap", None),
                 ("deep_gemm", "m_grouped_gemm_fp8_fp8_bf16_nt_offset"), ("eps.executor", "silu"), ("eps.fast_ep", "AllToAll"),
                 ("flashinfer.comm", None), ("flashinfer", "merge_state"), ("flashinfer", "dsv3_router_gemm"), ("flashinfer", "FusedSetKVBufferArg")}
    have = {(r["module"], r["name"]) for r in surface()["records"]}
    assert unguarded <= have, unguarded - have          # the fixture really records them
    for key in sorted(unguarded, key=str):
        resolve(*key)


def test_communicator_host_objects_carry_the_group(shim_first):
    """parallel_state.py:963-977: rank 0 draws an id, every rank builds the communicator; fast_ep.py:15-22 hands `data_ptr()` to AllToAll."""
    from eps.communication import MscclppCommunicator, MscclppCommunicatorParams
    from fluent_mi355.comm import communicator_from_ptr

    uid = MscclppCommunicator.createUniqueId()
    import pickle

    assert pickle.loads(pickle.dumps(uid)) == uid                      # travels through broadcast_object_list
    c = MscclppCommunicator(uid, MscclppCommunicatorParams(0, 1, 1))
    assert communicator_from_ptr(c.data_ptr()) is c and communicator_from_ptr(None) is None and communicator_from_ptr(0) is None
    assert (c.rank, c.world_size) == (0, 1)
    with pytest.raises(RuntimeError):
        communicator_from_ptr(c.data_ptr() + 1)
    with pytest.raises(ValueError):
        MscclppCommunicator("not-an-id", MscclppCommunicatorParams(0, 1, 1))
    with pytest.raises(RuntimeError):
        MscclppCommunicator(uid, MscclppCommunicatorParams(1, 2, 2))   # multi-rank without torch.distributed
    c.close()
    with pytest.raises(RuntimeError):
        communicator_from_ptr(c.data_ptr())
