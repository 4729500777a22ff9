#!/usr/bin/env python3
"""What a pure store stream and a 1 : 4 read : write stream (the shape of a 2x float up-scale) sustain on this device: torch's
fill and elementwise kernels over 2 GiB, timed with events.  Context for the float planes' kernels (C4 moves 2.69 GB per 1.03 ms)."""
import torch

dev = torch.device("cuda:0")
n = 512 * 1024 * 1024  # floats = 2 GiB
x = torch.empty(n, dtype=torch.float32, device=dev)
src = torch.rand(n // 4, dtype=torch.float32, device=dev)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


t = timed(lambda: x.fill_(1.5))
print(f"fill 2 GiB: {t * 1e3:.3f} ms = {x.numel() * 4 / t / 1e12:.2f} TB/s written")
y = x.view(4, n // 4)
t = timed(lambda: torch.add(src.unsqueeze(0), 1.0, out=y[:1]) if False else y.copy_(src.unsqueeze(0).expand(4, -1)))
print(f"read 0.5 GiB, write 2 GiB (broadcast copy): {t * 1e3:.3f} ms = {(x.numel() * 4 + src.numel() * 4) / t / 1e12:.2f} TB/s moved")
z = torch.empty_like(x)
t = timed(lambda: z.copy_(x))
print(f"copy 2 GiB -> 2 GiB: {t * 1e3:.3f} ms = {2 * x.numel() * 4 / t / 1e12:.2f} TB/s moved")
