"""Experiment: does grouping molecules of similar size into the same 64-graph unit speed the lane kernel up?
Times gtok_sent on the ZINC-shaped corpus as generated, and with the graphs reordered by node count."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 249456
d = gtok.synth.zinc_like(G, seed=1000)

def reorder(d, perm):
    nc, ec = d["node_counts"], d["edge_counts"]
    nptr = np.concatenate([[0], np.cumsum(nc)]); eptr = np.concatenate([[0], np.cumsum(ec)])
    nidx = np.concatenate([np.arange(nptr[g], nptr[g + 1]) for g in perm]) if len(perm) < 1000 else None
    # vectorised gather of variable-length segments
    def seg(ptr, cnt):
        c = cnt[perm]; start = ptr[perm]
        off = np.repeat(start - (np.cumsum(c) - c), c)
        return np.arange(int(c.sum())) + off
    ni, ei = seg(nptr[:-1], nc), seg(eptr[:-1], ec)
    return dict(node_counts=nc[perm], edge_counts=ec[perm], src=d["src"][ei], dst=d["dst"][ei], x=d["x"][ni], edge_attr=d["edge_attr"][ei])

def run(tag, d):
    host = gtok.GraphBatch.from_coo(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"])
    b = host.to(dev)
    ld = 200
    ids = torch.empty((G, ld), dtype=torch.int32, device=dev); ln = torch.empty(G, dtype=torch.int32, device=dev)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    for _ in range(3):
        gtok.ops.sent(b, 37, 1024, 0, 0, ld=ld, out=(ids, ln), **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(20):
        gtok.ops.sent(b, 37, 1024, 0, k, ld=ld, out=(ids, ln), **kw)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f"{tag:28s} chunk_nodes {host.chunk_nodes:5d} chunk_edges {host.chunk_edges:5d}  {ms:7.4f} ms  {G / ms / 1e3:8.1f} M graphs/s", flush=True)

run("as generated", d)
run("sorted by node count", reorder(d, np.argsort(d["node_counts"], kind="stable")))
run("sorted descending", reorder(d, np.argsort(-d["node_counts"], kind="stable")))
