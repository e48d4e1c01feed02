"""GPU box: on-box HBM ceilings to read the roofline fractions against: device-to-device copy (read + write)
and a read-only reduction over 4 GiB, torch kernels, median of 10."""
import torch, time
dev = torch.device("cuda:0")
x = torch.empty(1 << 30, dtype=torch.float32, device=dev).normal_()
y = torch.empty_like(x)
def med(fn, n=10):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
y.copy_(x); x.sum(); torch.cuda.synchronize()
t_copy = med(lambda: y.copy_(x))
t_sum = med(lambda: x.sum())
print(f"d2d copy 4 GiB: {2 * x.numel() * 4 / t_copy / 1e6:.0f} GB/s (read+write); read-only sum 4 GiB: {x.numel() * 4 / t_sum / 1e6:.0f} GB/s")
