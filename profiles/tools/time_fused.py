"""gtok_sent_packed against the plain walk into the same 16-bit unpadded rows (HIP events, median of 30), ZINC-full-shaped corpus.
python profiles/tools/time_fused.py [K ...]   (GTOK_LIB=<variant .so> times a profiling build)"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda:0")
G, ld = int(os.environ.get("G", 249456)), 176
d = gtok.synth.zinc_like(G, seed=1000)
b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)


def ev(f, n=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


for K in [int(a) for a in sys.argv[1:]] or [1, 16]:
    ids = torch.empty((K * G, ld), dtype=torch.int16, device=dev); ln = torch.empty(K * G, dtype=torch.int32, device=dev)
    walk = lambda: gtok.ops.sent(b, b.max_nodes, 1024, 0, 0, ld=ld, out=(ids, ln), pad=False, epochs=K, u16=True, **kw)
    pk = gtok.ops.PackedRows(K * G, K * G * 104, True, dev)
    fused = lambda: gtok.ops.sent(b, b.max_nodes, 1024, 0, 0, ld=ld, out=(ids, ln), pad=False, epochs=K, u16=True, packed=pk, **kw)
    for _ in range(200 if K == 1 else 20):
        walk()
    t_walk, t_fused = ev(walk), ev(fused)
    t_walk2 = ev(walk)
    print(f"K={K:2d} {os.path.basename(os.environ.get('GTOK_LIB', 'libgtok.so'))}: walk {t_walk:.4f} / {t_walk2:.4f} ms, walk that packs {t_fused:.4f} "
          f"(per epoch {t_walk / K:.5f} -> {t_fused / K:.5f}, +{(t_fused - t_walk) / K * 1e3:.2f} us); used {int(pk.used())} status {int(pk.status())} fused {pk.fused}")
    del ids, ln, pk
