"""Host -> device rates on this box: one pinned hipMemcpyAsync, pageable copy, and multi-threaded host memcpy into
pinned memory (what the staging step of lsr_recognize_host does)."""
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

n = 1 << 30
src = torch.empty(n, dtype=torch.uint8).pin_memory()
src.fill_(1)
dst = torch.empty(n, dtype=torch.uint8, device="cuda")
for _ in range(2):
    dst.copy_(src, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    dst.copy_(src, non_blocking=True)
torch.cuda.synchronize()
print("pinned H2D GB/s:", 5 * n / (time.perf_counter() - t0) / 1e9)
for chunk in (16 << 20, 64 << 20):
    t0 = time.perf_counter()
    for off in range(0, n, chunk):
        dst[off:off + chunk].copy_(src[off:off + chunk], non_blocking=True)
    torch.cuda.synchronize()
    print(f"pinned H2D in {chunk >> 20} MiB pieces GB/s:", n / (time.perf_counter() - t0) / 1e9)
page = np.ones(n, np.uint8)
pin = src.numpy()
for threads in (1, 4, 8, 16):
    piece = n // (threads * 8)
    with ThreadPoolExecutor(threads) as ex:
        def cp(i):
            pin[i * piece:(i + 1) * piece] = page[i * piece:(i + 1) * piece]
        list(ex.map(cp, range(threads * 8)))
        t0 = time.perf_counter()
        list(ex.map(cp, range(threads * 8)))
        print(f"host memcpy pageable->pinned, {threads} threads GB/s:", n / (time.perf_counter() - t0) / 1e9)
