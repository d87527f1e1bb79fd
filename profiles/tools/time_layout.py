"""One-off layout steps on a resident batch, timed on the stream (HIP events): gtok_csr_check, gtok_csr_lane_sort
(ops.lane_sorted), and the whole prepared_args call with its read-back.  python profiles/tools/time_layout.py [graphs]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda:0")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 249456
d = gtok.synth.zinc_like(G, seed=1000)
b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)
print("flags", b.flags, "max_degree", b.max_degree)


def ev(f, n=20):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def wall(f, n=10):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def sort_once():
    b.lane_sorted = None
    return gtok.ops.lane_sorted(b)


print("csr_check          event ms (median, min):", ev(lambda: gtok.ops.csr_check(b)))
print("lane_sorted        event ms (median, min):", ev(sort_once), " wall:", wall(sort_once))
raw = dict(node_ptr=b.node_ptr, edge_ptr=b.edge_ptr, rowptr=b.rowptr, col=b.col, nattr=b.nattr, eattr=b.eattr)
print("prepared_args wall ms (median, min):", wall(lambda: gtok.torch_ops.prepared_args(**raw, max_nodes=b.max_nodes, max_edges=b.max_edges)))
sb = sort_once()
print("units", sb.num_units, "chunk", sb.chunk_nodes, sb.chunk_edges)
