"""MI355X: R1 router selection (`flashinfer.moe_fused_gate`, csrc/moe_gate.hip) through the C-ABI against the golden vectors
of the reference's biased_grouped_topk_impl and against the oracle on larger seeded inputs."""
import numpy as np
import pytest
import torch

from helpers import load_golden
from test_oracle_golden import ROUTER_CASES, TOPK_PLAIN_CASES, assert_router_rows_equal, router_case, topk_plain_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def run_hip(logits, bias, G, TG, K, scale, on_out, npad):
    import flashinfer

    n = None if npad is None else torch.tensor(npad, dtype=torch.int32, device=DEV)
    w, ids = flashinfer.moe_fused_gate(torch.from_numpy(logits).to(DEV), torch.from_numpy(bias).to(DEV), G, TG, K, 0, scale,
                                       on_out, num_token_non_padded=n)
    torch.cuda.synchronize()
    return w.cpu().numpy(), ids.cpu().numpy()


@pytest.mark.parametrize("name", ROUTER_CASES)
def test_moe_fused_gate_vs_reference_golden(name):
    c = router_case(load_golden("router_biased_grouped_topk.npz"), name)
    w, ids = run_hip(c["logits"], c["bias"], c["G"], c["TG"], c["K"], c["scale"], c["on_out"], c["npad"])
    assert_router_rows_equal(w, ids, c["w"], c["ids"], c["npad"])


@pytest.mark.parametrize("T,E,G,TG,K", [(1, 256, 8, 4, 8), (4099, 256, 8, 4, 8), (257, 512, 16, 4, 12), (300, 64, 1, 1, 8),
                                        (33, 1024, 8, 3, 6), (64, 128, 64, 5, 5)])
def test_moe_fused_gate_vs_oracle(T, E, G, TG, K):
    """Seeded logits at decode and prefill token counts, every experts-per-lane instantiation (E/64 = 1..16); rows come out
    ordered by descending choice score, so ids are compared position by position too."""
    from oracle import router_ref

    rng = np.random.default_rng(T * 7 + E)
    logits = (rng.standard_normal((T, E)) * 3).astype(np.float32)
    bias = (rng.standard_normal(E) * 0.2).astype(np.float32)
    w_ref, ids_ref = router_ref.biased_grouped_topk(logits, bias, G, TG, K, 2.5, True)
    w, ids = run_hip(logits, bias, G, TG, K, 2.5, True, None)
    # float32 sigmoid on device vs float64-rounded on the host: a 1-ulp difference can swap two near-equal picks
    same = (ids == ids_ref).all(1)
    assert same.mean() > 0.999
    assert_router_rows_equal(w[same], ids[same], w_ref[same], ids_ref[same], None)
    assert np.allclose(w.sum(1), 2.5, rtol=1e-5)


def test_moe_fused_gate_empty_and_bad_arguments():
    import flashinfer

    w, ids = flashinfer.moe_fused_gate(torch.empty(0, 256, device=DEV), torch.zeros(256, device=DEV), 8, 4, 8, 0, 2.5, True)
    assert w.shape == (0, 8) and ids.shape == (0, 8)
    with pytest.raises(RuntimeError):
        flashinfer.moe_fused_gate(torch.zeros(2, 96, device=DEV), torch.zeros(96, device=DEV), 8, 4, 8, 0, 2.5, True)   # E not 64*2^k
    with pytest.raises(RuntimeError):
        flashinfer.moe_fused_gate(torch.zeros(2, 256, device=DEV), torch.zeros(256, device=DEV), 8, 1, 64, 0, 2.5, True)  # topk > kept experts
    with pytest.raises(NotImplementedError):
        flashinfer.moe_fused_gate(torch.zeros(2, 256, device=DEV), torch.zeros(256, device=DEV), 8, 4, 8, 1, 2.5, True)


# ---- R1b: flashinfer.topk_softmax / flashinfer.routing_flash / eps topk_sigmoid (csrc/moe_gate.hip: topk_gate_kernel) ----
@pytest.mark.parametrize("name", TOPK_PLAIN_CASES)
def test_plain_topk_routers_vs_reference_golden(name):
    """topk_softmax as fused_topk calls it (topk.py:513-518: outputs passed in) and routing_flash as select_experts calls it (:845),
    against the golden vectors of their torch statements run from the reference's source."""
    import flashinfer

    c = topk_plain_case(load_golden("router_topk_plain.npz"), name)
    x = torch.from_numpy(c["logits"]).to(DEV)
    T = x.shape[0]
    w = torch.empty(T, c["K"], dtype=torch.float32, device=DEV)
    ids = torch.empty(T, c["K"], dtype=torch.int32, device=DEV)
    if c["bias"] is None:
        flashinfer.topk_softmax(w, ids, x, c["renorm"])
    else:
        flashinfer.routing_flash(x, torch.from_numpy(c["bias"]).to(DEV), ids, w, x.shape[1], None, c["renorm"])
    torch.cuda.synchronize()
    assert_router_rows_equal(w.cpu().numpy(), ids.cpu().numpy(), c["w"], c["ids"], None)


@pytest.mark.parametrize("T,E,K", [(1, 8, 2), (513, 128, 8), (77, 768, 12), (5, 1000, 64), (64, 60, 4)])
def test_plain_topk_routers_vs_oracle(T, E, K):
    """Every experts-per-lane instantiation incl. expert counts that are not multiples of 64; bf16 logits and int64 ids (the dtypes
    topk_config.topk_indices_dtype may ask for); scaling; sigmoid scores (eps topk_sigmoid)."""
    import flashinfer
    from eps.utils.ops._ops import topk_sigmoid
    from oracle import router_ref

    rng = np.random.default_rng(T * 3 + E)
    logits = (rng.standard_normal((T, E)) * 3).astype(np.float32)
    bias = (rng.standard_normal(E) * 0.05).astype(np.float32)
    x = torch.from_numpy(logits).to(DEV)
    for mode in ("softmax", "flash", "sigmoid"):
        w = torch.empty(T, K, dtype=torch.float32, device=DEV)
        ids = torch.empty(T, K, dtype=torch.int64 if mode == "flash" else torch.int32, device=DEV)
        if mode == "softmax":
            flashinfer.topk_softmax(w, ids, x, True)
            w_ref, ids_ref = router_ref.topk_plain(logits, K, True)
        elif mode == "flash":
            flashinfer.routing_flash(x, torch.from_numpy(bias).to(DEV), ids, w, E, 2.5, False)
            w_ref, ids_ref = router_ref.topk_plain(logits, K, False, bias, scale=2.5)
        else:
            topk_sigmoid(w, ids, x, True)
            w_ref, ids_ref = router_ref.topk_plain(logits, K, True, sigmoid=True)
        torch.cuda.synchronize()
        got_w, got_ids = w.cpu().numpy(), ids.cpu().numpy().astype(np.int32)
        same = (got_ids == ids_ref).all(1)        # float32 exp on device vs float64-rounded on the host: near-ties may swap
        assert same.mean() > 0.99, mode
        assert np.abs(got_w[same] - w_ref[same]).max() < 5e-6 * max(1.0, float(np.abs(w_ref).max())), mode


def test_plain_topk_routers_reject_bad_arguments():
    import flashinfer

    x = torch.zeros(4, 128, device=DEV)
    with pytest.raises(RuntimeError):
        flashinfer.topk_softmax(torch.empty(4, 8, device=DEV), torch.empty(4, 8, dtype=torch.int16, device=DEV), x, True)
    with pytest.raises(RuntimeError):
        flashinfer.topk_softmax(torch.empty(3, 8, device=DEV), torch.empty(3, 8, dtype=torch.int32, device=DEV), x, True)
    with pytest.raises(RuntimeError):
        flashinfer.topk_softmax(torch.empty(4, 200, device=DEV), torch.empty(4, 200, dtype=torch.int32, device=DEV), x, True)   # topk > E
    w, ids = torch.empty(0, 8, device=DEV), torch.empty(0, 8, dtype=torch.int32, device=DEV)
    flashinfer.topk_softmax(w, ids, torch.empty(0, 128, device=DEV), True)   # empty batch: nothing to do
