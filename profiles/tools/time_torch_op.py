"""Where a torch.ops.gtok.sent call spends its time: the same ZINC-full epoch through ops.sent (caller's buffers / fresh
buffers) and through the custom op (raw tensors / prepared arrays), one HIP event pair around N launches; and the host cost
per call alone, on a batch whose kernel is short.  python profiles/tools/time_torch_op.py"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda:0")


def run(G, N):
    d = gtok.synth.zinc_like(G, seed=1000)
    b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)
    ld, mn = 176, b.max_nodes
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    ids = torch.empty((G, ld), dtype=torch.int32, device=dev); ln = torch.empty(G, dtype=torch.int32, device=dev)
    raw = dict(node_ptr=b.node_ptr, edge_ptr=b.edge_ptr, rowptr=b.rowptr, col=b.col, nattr=b.nattr, eattr=b.eattr)
    P = gtok.torch_ops.prepared_args(**raw, max_nodes=b.max_nodes, max_edges=b.max_edges)
    opkw = dict(query=None, max_num_nodes=mn, max_len=1024, ld=ld, seed=0, labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True, pad_id=5, graph_base=0)
    rawkw = dict(raw, max_nodes=b.max_nodes, max_edges=b.max_edges)
    variants = {
        "ops.sent, caller's buffers": lambda k: gtok.ops.sent(b, mn, 1024, 0, k, ld=ld, out=(ids, ln), **kw),
        "ops.sent, fresh buffers": lambda k: gtok.ops.sent(b, mn, 1024, 0, k, ld=ld, **kw),
        "torch op, raw tensors": lambda k: torch.ops.gtok.sent(**rawkw, epoch=k, **opkw),
        "torch op, prepared": lambda k: torch.ops.gtok.sent(**P, epoch=k, **opkw),
    }
    for rep in range(2):
        for name, f in variants.items():
            for k in range(10):
                f(k)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter(); s.record()
            for k in range(N):
                f(k)
            t_issue = time.perf_counter() - t0
            e.record(); torch.cuda.synchronize()
            print(f"G={G:7d} {name:30s} stream ms/call {s.elapsed_time(e) / N:.5f}   host issue us/call {t_issue / N * 1e6:7.2f}   kernel {gtok.ops.last_sent_kernel()}")


run(249456, 300)
run(30000, 2000)
