import torch, time
dev = "cuda"
for logn in (24, 26, 28, 30):
    n = 1 << logn
    x = torch.arange(n, dtype=torch.int32, device=dev)
    m = 1 << 27
    idx = torch.randint(0, n, (m,), device=dev, dtype=torch.int64)
    for width in (1, 16):
        if width == 1:
            f = lambda: x[idx]
        else:
            xv = x.view(-1, 16)
            iv = idx[: m // 4] % (n // 16)
            f = lambda: xv[iv]
        f(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3): f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 3
        cnt = m if width == 1 else m // 4
        print("table %5d MB, gather of %2d B rows: %.2f G rows/s (%.1f GB/s useful)" % (n * 4 >> 20, 4 * width, cnt / dt / 1e9, cnt * 4 * width / dt / 1e9))
