"""GPU: the driver's contract with bench.py — ONE JSON line with the agreed keys — and the multi-GPU code path that cannot run on a
1-GPU box otherwise: FLUENT_BENCH_CFG4=force makes the N = 1 run go through the config-4 CHILD-process phase of an N > 1 run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys_and_the_config4_child_record():
    env = dict(os.environ, FLUENT_BENCH_CFG4="force")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-gemm"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 128 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3       # tokens/s = bs / step time
    assert d["config"]["quant_launch"] == "K5, K4 separate" and d["config"]["launches_per_layer"] == 3    # the reference's call sequence
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / roof["us_per_launch"] / 1e3) / roof["achieved"] < 1e-2
    # the compute-side fraction and the chip's clock / power during the same launches ride in the record
    for k in ("mfma_frac", "binding", "sclk_mhz_mean", "power_w_mean", "power_cap_w"):
        assert k in roof, k
    assert roof["binding"] in ("hbm", "issue/power")
    flops = 2176.0 * 128 * 4096 * 128
    assert abs(roof["mfma_frac"] - flops / (roof["us_per_launch"] * 1e-6) / 5e15) < 2e-3
    assert roof["sclk_mhz_mean"] is None or 300 <= roof["sclk_mhz_mean"] <= 3000
    assert roof["power_w_mean"] is None or 50 <= roof["power_w_mean"] <= 2500
    step = (d.get("variants") or {}).get("step")
    if step is not None:                                             # (round 6) K3 alone rides in the record: it is on the host-critical path of a replay
        assert 0 < step["k3_us_per_call"] < 200
    c4 = d["cfg4"]
    assert c4 is not None and "error" not in c4, c4
    assert c4["value"] > 0 and c4["scaling"] == "strong" and c4["config"]["data_connected"] is True and "attempt" in c4
