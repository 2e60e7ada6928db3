"""What a plain streaming kernel gets out of the HBM on this box: device-to-device copies moving the forward gather's
algorithmic byte count (973.5 MB per launch) and a few other sizes; GB/s = (bytes read + bytes written) / time."""
import torch
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for total_mb in (973.5, 2000.0, 8000.0):
    n = int(total_mb * 1e6 / 2 / 4)
    src = torch.randn(n, device='cuda'); dst = torch.empty_like(src)
    us = t(lambda: dst.copy_(src))
    print('copy moving %.1f MB (read + write): %.1f us = %.0f GB/s = %.3f of 8 TB/s' % (total_mb, us, total_mb * 1e6 / us / 1e3, total_mb * 1e6 / us / 1e3 / 8000))
    wus = t(lambda: dst.zero_())
    print('  fill of %.1f MB: %.1f us = %.0f GB/s' % (total_mb / 2, wus, total_mb / 2 * 1e6 / wus / 1e3))
    del src, dst
