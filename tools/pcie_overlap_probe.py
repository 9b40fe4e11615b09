"""Probe: do a pinned H2D and a pinned D2H copy on two HIP streams overlap (full-duplex PCIe)?"""
import time
import torch

n = 64 << 20
h_in = torch.empty(n, dtype=torch.uint8, pin_memory=True)
h_out = torch.empty(n, dtype=torch.uint8, pin_memory=True)
d_a = torch.empty(n, dtype=torch.uint8, device="cuda")
d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def t(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def h2d():
    with torch.cuda.stream(s1):
        d_a.copy_(h_in, non_blocking=True)


def d2h():
    with torch.cuda.stream(s2):
        h_out.copy_(d_b, non_blocking=True)


def both():
    h2d()
    d2h()


a, b, c = t(h2d), t(d2h), t(both)
print(f"H2D {n / a / 1e9:.1f} GB/s  D2H {n / b / 1e9:.1f} GB/s  both at once: {c * 1e3:.2f} ms vs {a * 1e3:.2f} + {b * 1e3:.2f} ms "
      f"(overlap {'yes' if c < 0.8 * (a + b) else 'NO'})")
