import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
d = gtok.synth.zinc_like(249456, seed=1000)
host = gtok.GraphBatch.from_coo(d["node_counts"], d["edge_counts"], d["src"], d["dst"])
for pin in ("lane", "blane", "reg"):
    os.environ["GTOK_SENT_KERNEL"] = pin
    b = host.to(dev)
    ids = torch.empty((b.num_graphs, 96), dtype=torch.int32, device=dev); ln = torch.empty(b.num_graphs, dtype=torch.int32, device=dev)
    for _ in range(3): gtok.ops.sent(b, 37, 1024, 0, 0, ld=96, out=(ids, ln))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(20): gtok.ops.sent(b, 37, 1024, 0, k, ld=96, out=(ids, ln))
    e.record(); torch.cuda.synchronize()
    print(pin, gtok.ops.sent_kernel_name(b, 37, 1024), f"{s.elapsed_time(e)/20:.4f} ms", "maxlen", int(ln.max()), flush=True)
