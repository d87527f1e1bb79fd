"""Where does sent_lane_kernel (reordered batch) overtake the wave-per-graph sent_reg_kernel on ZINC-shaped molecules?"""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
for G in (8000, 12000, 16000, 20000, 24000, 28000, 32000, 48000):
    d = gtok.synth.zinc_like(G, seed=1000)
    host = gtok.GraphBatch.from_coo(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"])
    res = {}
    for pin in ("reg", "lane"):
        os.environ["GTOK_SENT_KERNEL"] = pin
        b = host.to(dev)
        ids = torch.empty((G, 208), dtype=torch.int32, device=dev); ln = torch.empty(G, dtype=torch.int32, device=dev)
        for _ in range(3): gtok.ops.sent(b, 37, 1024, 0, 0, ld=208, out=(ids, ln), **kw)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for k in range(20): gtok.ops.sent(b, 37, 1024, 0, k, ld=208, out=(ids, ln), **kw)
        e.record(); torch.cuda.synchronize()
        res[pin] = s.elapsed_time(e) / 20
    print(f"G={G}: reg {res['reg']:.4f} ms  lane {res['lane']:.4f} ms", flush=True)
