"""Measure plain device fill / copy bandwidth (context for the write-dominated first layer)."""
import torch
n = 64 * 513 * 256 * 64          # floats the first layer writes per launch (batch 64)
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
t = timeit(lambda: x.fill_(1.0)); print(f"fill  {n*4/1e9:.2f} GB: {t:.3f} ms  {n*4/t/1e9:.2f} TB/s written")
t = timeit(lambda: y.copy_(x));   print(f"copy  {n*4/1e9:.2f} GB: {t:.3f} ms  {2*n*4/t/1e9:.2f} TB/s moved")
t = timeit(lambda: torch.relu_(x)); print(f"relu_ {n*4/1e9:.2f} GB: {t:.3f} ms  {2*n*4/t/1e9:.2f} TB/s moved")
