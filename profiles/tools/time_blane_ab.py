"""sent_blane_kernel<4,8> on the config-5 share (125 k ER graphs of 10-256 nodes, max_len 600): padded and GTOK_SENT_NO_PAD,
best of 3 x 40 back-to-back launches."""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
G = 125000
d = gtok.synth.er_batch_device(G, dev, seed=1000)
b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], device=dev)
ids = torch.empty((G, 608), dtype=torch.int32, device=dev); ln = torch.empty(G, dtype=torch.int32, device=dev)
res = []
for pad in (True, False):
    for _ in range(5): gtok.ops.sent(b, 256, 600, 0, 0, ld=608, out=(ids, ln), pad=pad)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for k in range(40): gtok.ops.sent(b, 256, 600, 0, k, ld=608, out=(ids, ln), pad=pad)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 40)
    res.append(f"{'padded' if pad else 'no_pad'} {best:.4f} ms")
print(os.path.basename(gtok._lib.LIB_PATH), gtok.ops.sent_kernel_name(b, 256, 600), " | ".join(res), f"(checksum {int(ln.sum())})", flush=True)
