"""Times gtok_sent on config-5 shaped batches (unlabelled, 10-256 nodes) under each applicable kernel pin.
usage: python profiles/tools/time_sent_large.py [G]"""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 125000


def run(name, d, pin, order="1"):
    os.environ["GTOK_SENT_KERNEL"] = pin
    os.environ["GTOK_BLANE_ORDER"] = order
    b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], device=dev)
    ids = torch.empty((b.num_graphs, 608), dtype=torch.int32, device=dev); ln = torch.empty(b.num_graphs, dtype=torch.int32, device=dev)
    for _ in range(3):
        gtok.ops.sent(b, b.max_nodes, 600, 0, 0, ld=608, out=(ids, ln))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(10):
        gtok.ops.sent(b, b.max_nodes, 600, 0, k, ld=608, out=(ids, ln))
    e.record(); torch.cuda.synchronize()
    print(f"{name:6s} G={b.num_graphs} {gtok.ops.sent_kernel_name(b, b.max_nodes, 600):24s} order={order}: {s.elapsed_time(e) / 10:.4f} ms", flush=True)
    return ids.clone(), ln.clone()


for name, d in (("er", gtok.synth.er_batch_device(G, dev, seed=1000)), ("mix", gtok.synth.mix_batch_device(G, dev, seed=1000))):
    a = run(name, d, "lds")
    for order in ("1", "0"):
        c = run(name, d, "blane", order)
        assert torch.equal(a[1], c[1]) and torch.equal(a[0], c[0]), "kernels disagree"
