"""Order of the (unit, epoch) pairs of a K-epoch launch (GTOK_LANE_PAIR_ORDER=unit|epoch), corpora x K x row flavours."""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
for G, Ks in ((249456, (2, 4)), (31182, (16, 32)), (12000, (24, 48))):
    d = gtok.synth.zinc_like(G, seed=1000)
    b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)
    for K in Ks:
        for u16, pad in ((False, True), (True, False)):
            ids = torch.empty((K * G, 208), dtype=torch.int16 if u16 else torch.int32, device=dev)
            ln = torch.empty(K * G, dtype=torch.int32, device=dev)
            res = {}
            for rep in range(2):
                for order in ("unit", "epoch"):
                    os.environ["GTOK_LANE_PAIR_ORDER"] = order
                    f = lambda k: gtok.ops.sent(b, 37, 1024, 0, k * K, ld=208, out=(ids, ln), pad=pad, epochs=K, u16=u16, **kw)
                    for _ in range(5): f(0)
                    torch.cuda.synchronize()
                    n = max(10, 200000 // (G * K) * 4)
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for k in range(n): f(k)
                    e.record(); torch.cuda.synchronize()
                    res.setdefault(order, []).append(s.elapsed_time(e) / n / K)
            print(f"G={G} K={K} {'u16 nopad' if u16 else 'i32 pad  '}: per epoch unit-major {min(res['unit']):.5f} ({max(res['unit']):.5f})  epoch-major {min(res['epoch']):.5f} ({max(res['epoch']):.5f})", flush=True)
            del ids, ln
