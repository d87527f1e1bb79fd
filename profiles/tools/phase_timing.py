"""Per-phase cycle counts of sent_lds_kernel on the large-graph corpus.  Needs a library built with
-DGTOK_PHASE_TIMING (profiling build: the last 4 columns of every row then hold s_memtime deltas for
stage+zero+draws / adjacency build / walk / row write).  Never ship that build."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
max_len = int(sys.argv[2]) if len(sys.argv) > 2 else 600
d = gtok.synth.er_batch_device(G, dev, seed=1000)
host = gtok.GraphBatch.from_coo(d["node_counts"], d["edge_counts"], d["src"], d["dst"])
b = host.to(dev)
ld = max_len + 8
for k in range(3):
    ids, ln = gtok.ops.sent(b, host.max_nodes, max_len, 0, k, ld=ld)
torch.cuda.synchronize()
ph = ids[:, -4:].double()
names = ["stage+zero+draws", "adjacency build", "walk", "row write"]
tot = ph.sum()
for i, nme in enumerate(names):
    print(f"{nme:18s} mean {float(ph[:, i].mean()):10.0f} ticks  max {float(ph[:, i].max()):10.0f}  share {float(ph[:, i].sum() / tot):.3f}")
print("mean ticks per graph", float(ph.sum(1).mean()), "(s_memtime ticks: 100 MHz)")

