import torch, time
dev = torch.device("cuda:0")
for mb in (1, 4, 7.2, 16, 64):
    n = int(mb * 1e6)
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    dst = torch.empty(n, dtype=torch.uint8).pin_memory()
    for _ in range(3):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print("D2H %.1f MB: median %.1f us = %.1f GB/s" % (mb, ts[5] * 1e6, n / ts[5] / 1e9))
