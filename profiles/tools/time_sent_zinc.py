"""Time gtok_sent on the ZINC-shaped corpus (labelled and unlabelled) under the kernel pinned by GTOK_SENT_KERNEL."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 249456
d = gtok.synth.zinc_like(G, seed=1000)
for labeled in (True, False):
    host = gtok.GraphBatch.from_coo(d["node_counts"], d["edge_counts"], d["src"], d["dst"],
                                    d["x"] if labeled else None, d["edge_attr"] if labeled else None)
    b = host.to(dev)
    ld = 208 if labeled else 128
    ids = torch.empty((G, ld), dtype=torch.int32, device=dev); ln = torch.empty(G, dtype=torch.int32, device=dev)
    kw = dict(labeled=labeled, num_node_types=9 if labeled else 0, num_edge_types=4 if labeled else 0, remap_zinc=labeled)
    for _ in range(3):
        gtok.ops.sent(b, 37, 1024, 0, 0, ld=ld, out=(ids, ln), **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(20):
        gtok.ops.sent(b, 37, 1024, 0, k, ld=ld, out=(ids, ln), **kw)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f"{os.environ.get('GTOK_SENT_KERNEL','auto'):5s} labeled={labeled!s:5s} {ms:7.4f} ms  {G / ms / 1e3:8.1f} M graphs/s  max len {int(ln.max())}")
