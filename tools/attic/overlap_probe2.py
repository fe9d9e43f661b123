#!/usr/bin/env python3
import time, torch
pin = torch.empty(360 << 20, dtype=torch.uint8).pin_memory()
dst = torch.empty(360 << 20, dtype=torch.uint8, device="cuda")
a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
side = torch.cuda.Stream()
small_pin = torch.empty(5 << 20, dtype=torch.uint8).pin_memory(); small_dev = torch.empty(5 << 20, dtype=torch.uint8, device="cuda")
def copy():
    with torch.cuda.stream(side): dst.copy_(pin, non_blocking=True)
def kern(n=12):
    x = a
    for _ in range(n): x = x @ a
    return x
def kern_d2h():
    kern(6); small_pin.copy_(small_dev, non_blocking=True); torch.cuda.current_stream().synchronize(); kern(6)
def t(fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(2): kern(); copy(); torch.cuda.synchronize()
print("kernels alone %.2f ms; copy alone %.2f ms; copy || kernels %.2f ms" % (t(kern), t(lambda: (copy(), side.synchronize())), t(lambda: (copy(), kern(), side.synchronize()))))
print("kernels + 5 MB D2H in the middle: alone %.2f ms; with the big copy %.2f ms" % (t(kern_d2h), t(lambda: (copy(), kern_d2h(), side.synchronize()))))
