import importlib, os, sys, torch
sys.path.insert(0, "/root/repo") if os.path.exists("/root/repo/bench.py") else None
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
G = 32768
d = gtok.synth.er_batch_device(G, dev, seed=1000)
b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], device=dev)
ids = torch.empty((G, 608), dtype=torch.int32, device=dev); ln = torch.empty(G, dtype=torch.int32, device=dev)
for _ in range(3): gtok.ops.sent(b, b.max_nodes, 600, 0, 0, ld=608, out=(ids, ln))
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for k in range(10): gtok.ops.sent(b, b.max_nodes, 600, 0, k, ld=608, out=(ids, ln))
e.record(); torch.cuda.synchronize()
print(f"lds kernel 32k ER graphs: {s.elapsed_time(e)/10:.4f} ms")
