"""Write-only and copy bandwidth of the box as torch sees it (fill_ / copy_ on 512 MiB): the practical ceiling for the
write-only generators (jakes_generate, randn_c)."""
import torch
x = torch.empty(1 << 27, dtype=torch.float32, device="cuda")      # 512 MiB
y = torch.empty_like(x)
def t(fn, nbytes, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print("%s: %.3f ms, %.0f GB/s (%.2f of 8 TB/s)" % (name, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 8000))
t(lambda: x.fill_(1.5), x.numel() * 4, "fill_ 512 MiB (write only)")
t(lambda: y.copy_(x), 2 * x.numel() * 4, "copy_ 512 MiB (read + write)")
t(lambda: x.sum(), x.numel() * 4, "sum 512 MiB (read only)")
