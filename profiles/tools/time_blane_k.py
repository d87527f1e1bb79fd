"""sent_blane_kernel at the dataset classes' epochs per launch (125 k ER graphs of 10-256 nodes, 16-bit rows, no padding): one number per
setting of GTOK_BLANE_WAVES / GTOK_BLANE_PRIO (printed)."""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
G, ld = 125000, 608
d = gtok.synth.er_batch_device(G, dev, seed=1000)
b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], device=dev)
out = []
for K in (1, 14):
    ids = torch.empty((K * G, ld), dtype=torch.int16, device=dev); ln = torch.empty(K * G, dtype=torch.int32, device=dev)
    f = lambda k: gtok.ops.sent(b, 256, 600, 0, k * K, ld=ld, out=(ids, ln), epochs=K, u16=True, pad=False)
    for _ in range(2): f(0)
    torch.cuda.synchronize(); best = 1e9
    n = 20 if K == 1 else 4
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for k in range(n): f(k)
        e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e) / n / K)
    out.append(f"K={K}: {best:.4f} ms per epoch")
    del ids, ln
print(f"WAVES={os.environ.get('GTOK_BLANE_WAVES', 'default')} PRIO={os.environ.get('GTOK_BLANE_PRIO', 'default')}  " + "  ".join(out), flush=True)
