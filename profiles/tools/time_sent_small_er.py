"""SENT on the reference's own synthetic shape (docs/synthetic_data.md:132-138): ER graphs of 10-49 nodes, sparsity
0.1-0.2, one direction per edge (u < v), unlabelled - the wave-per-graph register kernel's home ground."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
d = gtok.synth.er_batch_device(G, dev, seed=7, min_nodes=10, max_nodes=49)
b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], device=dev)
print("graphs", G, "nodes", b.num_nodes_total, "entries", b.num_edges_total, "flags", b.flags, "kernel", gtok.ops.sent_kernel_name(b, 49, 600))
ld = 600
ids = torch.empty((G, ld), dtype=torch.int32, device=dev); ln = torch.empty(G, dtype=torch.int32, device=dev)
for _ in range(3):
    gtok.ops.sent(b, 49, 600, 0, 0, ld=ld, out=(ids, ln))
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for k in range(10):
    gtok.ops.sent(b, 49, 600, 0, k, ld=ld, out=(ids, ln))
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print(f"{ms:.4f} ms  {G / ms / 1e3:.1f} M graphs/s  avg len {float(ln.float().mean()):.1f}  tokens/s {float(ln.sum()) / ms / 1e6:.1f} G")
