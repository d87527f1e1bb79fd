"""Reference points for the HBM roofline on this box: time torch's own fill / copy kernels on a slab of the
bench's size (240 MB int32) - what a plain streaming kernel reaches, against the 8 TB/s spec peak."""
import torch
dev = torch.device("cuda", 0)
G, ld = 249456, 240
a = torch.empty((G, ld), dtype=torch.int32, device=dev); b = torch.empty_like(a)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
nbytes = a.numel() * 4
ms = t(lambda: a.fill_(2)); print(f"fill  {nbytes/1e6:.0f} MB: {ms:.4f} ms  {nbytes/ms/1e6:.0f} GB/s written")
ms = t(lambda: b.copy_(a)); print(f"copy  {nbytes/1e6:.0f} MB: {ms:.4f} ms  {2*nbytes/ms/1e6:.0f} GB/s read+written")
