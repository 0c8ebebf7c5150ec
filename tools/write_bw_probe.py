#!/usr/bin/env python3
"""What a pure store stream reaches on this GPU: torch fill_ / zero_ / copy_ over buffers of configs[4]'s output size
(246 MB, rotating over three so that nothing stays in the 256 MB Infinity Cache) -- the practical ceiling the fused C5
evaluation (one 246 MB dense Jacobian per launch) is measured against."""
import torch, time
dev = "cuda:0"
N = 1024 * 200 * 150           # doubles: configs[4]'s dense Jacobian
bufs = [torch.empty(N, dtype=torch.float64, device=dev) for _ in range(3)]
src = torch.ones(N, dtype=torch.float64, device=dev)
def timed(fn, reps=60):
    for _ in range(6): fn(0); fn(1); fn(2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i % 3)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps
for name, fn, bytes_ in (("fill_", lambda i: bufs[i].fill_(1.5), N * 8), ("zero_", lambda i: bufs[i].zero_(), N * 8),
                         ("copy_ (read + write)", lambda i: bufs[i].copy_(src), 2 * N * 8)):
    t = timed(fn)
    print(f"{name:22s} {t*1e6:8.1f} us  {bytes_/t/1e12:6.2f} TB/s ({N*8/t/1e12:5.2f} TB/s of stores)")
