"""GPU box: is the compute-regime grouped GEMM power-limited?  Runs the w13 GEMM of BASELINE config 3 (T = 16384) in a loop for a
few seconds per variant while `rocm-smi` samples socket power and shader clock, for (a) the path's own random operands and
(b) all-zero operands (no data toggling: if the chip is at its power cap, zeros clock higher and run faster at the same
instruction stream).  usage: [FLUENT_GEMM_BIG=0] python tools/gemm_power_probe.py   (0: the 128-row tiles; default: the 256 x 256 kernel)"""
import json, os, subprocess, sys, threading, time, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-fluentllm_amd"))
import torch
import deep_gemm

dev = torch.device("cuda:0")
E, N, K, R = 256, 4096, 7168, 512
g = torch.Generator(device=dev).manual_seed(0)


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            pw = re.findall(r"Power[^:]*:\s*([0-9.]+)", r)
            sc = re.findall(r"sclk clock level[^(]*\((\d+)Mhz\)", r)
            out.append((float(pw[0]) if pw else None, int(sc[0]) if sc else None))
        except Exception as ex:   # noqa
            out.append((None, None))
        time.sleep(0.1)


def run(tag, W, A):
    Ws = torch.rand(E, N // 128, K // 128, device=dev, generator=g) * 1e-2
    M = E * R
    As = torch.rand(M, K // 128, device=dev, generator=g)
    ex = (torch.arange(E + 1, device=dev) * R).to(torch.int32)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    f = lambda: deep_gemm.m_grouped_gemm_fp8_fp8_bf16_nt_offset((A, As), (W, Ws), out, ex, use_pdl=True)
    for _ in range(3): f()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, samples)); th.start()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 4.0:
        for _ in range(10): f()
        n += 10
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    ms = e0.elapsed_time(e1) / n
    pw = [p for p, _ in samples if p]; sc = [s for _, s in samples if s]
    print(json.dumps({"variant": tag, "big": "128-row tiles" if os.environ.get("FLUENT_GEMM_BIG") == "0" else "256 x 256 tiles (default)", "ms": round(ms, 3),
                      "TFLOPs": round(2.0 * M * N * K / ms / 1e9, 1), "power_W_mean": round(sum(pw) / max(len(pw), 1), 1),
                      "power_W_max": max(pw) if pw else None, "sclk_MHz_mean": round(sum(sc) / max(len(sc), 1)) if sc else None,
                      "samples": len(samples)}), flush=True)


M = E * R
Wr = torch.randint(0, 255, (E, N, K), device=dev, generator=g, dtype=torch.int16)
Wr = torch.where((Wr & 0x7F) == 0x7F, Wr - 1, Wr).to(torch.uint8).view(torch.float8_e4m3fn)
Ar = torch.randint(0, 255, (M, K), device=dev, generator=g, dtype=torch.int16)
Ar = torch.where((Ar & 0x7F) == 0x7F, Ar - 1, Ar).to(torch.uint8).view(torch.float8_e4m3fn)
run("random bytes", Wr, Ar)
run("zeros", torch.zeros_like(Wr), torch.zeros_like(Ar))
