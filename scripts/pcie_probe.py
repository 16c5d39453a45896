import torch, time
n = 300 * 1024 * 1024
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
h2 = torch.empty(80 * 1024 * 1024, dtype=torch.uint8).pin_memory()
d2 = torch.empty(80 * 1024 * 1024, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for both in (False, True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        with torch.cuda.stream(s1):
            d.copy_(h, non_blocking=True)
        if both:
            with torch.cuda.stream(s2):
                h2.copy_(d2, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("H2D %.1f GB/s%s" % (10 * n / dt / 1e9, " (with D2H %.1f GB/s beside)" % (10 * h2.numel() / dt / 1e9) if both else ""))
