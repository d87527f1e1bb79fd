"""Would sent_blane_kernel gain from a size-class split?  The config-5 share (125 k ER graphs, 10-256 nodes) as one W = 4 launch
against its graphs of <= 128 nodes (W = 2: 16 waves per CU) and of > 128 nodes (W = 4: 8 waves per CU) as two launches."""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
def run(G, lo, hi, seed):
    d = gtok.synth.er_batch_device(G, dev, seed=seed, min_nodes=lo, max_nodes=hi)
    b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], device=dev)
    ids = torch.empty((G, 608), dtype=torch.int32, device=dev); ln = torch.empty(G, dtype=torch.int32, device=dev)
    for _ in range(5): gtok.ops.sent(b, 256, 600, 0, 0, ld=608, out=(ids, ln))
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for k in range(40): gtok.ops.sent(b, 256, 600, 0, k, ld=608, out=(ids, ln))
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 40)
    print(f"{G} graphs of {lo}-{hi} nodes: {gtok.ops.sent_kernel_name(b, 256, 600)} {best:.4f} ms, avg len {float(ln.float().mean()):.0f}", flush=True)
    return best
full = run(125000, 10, 256, 1000)
small = run(int(125000 * 119 / 247), 10, 128, 1001)
large = run(125000 - int(125000 * 119 / 247), 129, 256, 1002)
print(f"one launch {full:.4f} ms, two launches {small + large:.4f} ms")
