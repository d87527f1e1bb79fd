"""What would sorting the corpus by molecule size buy sent_lane_kernel?  Times the ZINC-shaped corpus in dataset order
and physically sorted by node count (descending / ascending)."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
d = gtok.synth.zinc_like(249456, seed=1000)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)


def permuted(d, perm):
    nptr = np.concatenate([[0], np.cumsum(d["node_counts"])]); eptr = np.concatenate([[0], np.cumsum(d["edge_counts"])])
    nidx = np.concatenate([np.arange(nptr[g], nptr[g + 1]) for g in perm])
    eidx = np.concatenate([np.arange(eptr[g], eptr[g + 1]) for g in perm])
    return dict(node_counts=d["node_counts"][perm], edge_counts=d["edge_counts"][perm], src=d["src"][eidx], dst=d["dst"][eidx],
                x=d["x"][nidx], edge_attr=d["edge_attr"][eidx])


for name, perm in (("dataset order", None), ("descending size", np.argsort(-d["node_counts"], kind="stable")),
                   ("ascending size", np.argsort(d["node_counts"], kind="stable"))):
    dd = d if perm is None else permuted(d, perm)
    host = gtok.GraphBatch.from_coo(dd["node_counts"], dd["edge_counts"], dd["src"], dd["dst"], dd["x"], dd["edge_attr"])
    b = host.to(dev)
    ids = torch.empty((b.num_graphs, 208), dtype=torch.int32, device=dev); ln = torch.empty(b.num_graphs, dtype=torch.int32, device=dev)
    for _ in range(3): gtok.ops.sent(b, 37, 1024, 0, 0, ld=208, out=(ids, ln), **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(20): gtok.ops.sent(b, 37, 1024, 0, k, ld=208, out=(ids, ln), **kw)
    e.record(); torch.cuda.synchronize()
    print(f"{name:16s} {gtok.ops.sent_kernel_name(b, 37, 1024, **kw)} chunk_nodes {b.chunk_nodes} chunk_edges {b.chunk_edges}: {s.elapsed_time(e)/20:.4f} ms", flush=True)
