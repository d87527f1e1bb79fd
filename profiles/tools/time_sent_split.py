"""Time gtok_sent on the large-graph corpus at several max_len: max_len=2 is (almost) load+build only."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
d = gtok.synth.er_batch_device(G, dev, seed=1000)
host = gtok.GraphBatch.from_coo(d["node_counts"], d["edge_counts"], d["src"], d["dst"])
b = host.to(dev)
print("graphs", G, "nodes", host.num_nodes_total, "entries", host.num_edges_total, "max_nodes", host.max_nodes, "max_edges", host.max_edges)
for max_len in (2, 64, 200, 600, 4096):
    ld = (min(max_len, 2 + 5 * host.max_nodes + host.max_edges) + 3) // 4 * 4
    ids = torch.empty((G, ld), dtype=torch.int32, device=dev); ln = torch.empty(G, dtype=torch.int32, device=dev)
    for _ in range(3):
        gtok.ops.sent(b, host.max_nodes, max_len, 0, 0, ld=ld, out=(ids, ln))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(10):
        gtok.ops.sent(b, host.max_nodes, max_len, 0, k, ld=ld, out=(ids, ln))
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"max_len {max_len:5d} ld {ld:5d}  {ms:8.4f} ms  {G / ms / 1e3:8.2f} M graphs/s  avg len {float(ln.float().mean()):7.1f}")
