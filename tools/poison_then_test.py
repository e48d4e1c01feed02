"""GPU box: chase allocator-state-dependent failures — fill the caching allocator's free blocks with NaN-pattern bytes
(0x7F: fp8 NaN, 0x7F7F bf16 / 0x7F7F7F7F f32 NaN-ish), then run the selected tests in the SAME process, repeatedly."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
import pytest, torch
sel = sys.argv[1] if len(sys.argv) > 1 else "test_decode_parity_vs_oracle"
fails = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    blocks = [torch.full(((1 << 20) * s,), 0x7F, dtype=torch.uint8, device="cuda:0") for s in (1, 2, 8, 32, 128, 512, 2048)]
    small = [torch.full((n,), 0x7F, dtype=torch.uint8, device="cuda:0") for n in (512, 4096, 65536, 262144) for _ in range(64)]
    torch.cuda.synchronize()
    del blocks, small                      # back to the caching allocator, contents intact
    rc = pytest.main(["tests/test_mla_gpu.py", "-x", "-q", "-m", "gpu", "-k", sel, "-p", "no:cacheprovider"])
    fails += int(rc != 0)
    print(f"iteration {it}: rc={rc}", flush=True)
print("failures:", fails)
