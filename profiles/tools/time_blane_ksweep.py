import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
G, ld = 125000, 608
for kind in ("er", "mix"):
    d = gtok.synth.er_batch_device(G, dev, seed=1000) if kind == "er" else gtok.synth.mix_batch_device(G, dev, seed=1000)
    b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], device=dev)
    out = []
    for K in (8, 14, 20, 28):
        ids = torch.empty((K * G, ld), dtype=torch.int16, device=dev); ln = torch.empty(K * G, dtype=torch.int32, device=dev)
        f = lambda k: gtok.ops.sent(b, 256, 600, 0, k * K, ld=ld, out=(ids, ln), epochs=K, u16=True, pad=False)
        for _ in range(2): f(0)
        torch.cuda.synchronize(); best = 1e9
        for rep in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for k in range(3): f(k)
            e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e) / 3 / K)
        out.append(f"K={K}: {best:.4f}")
        del ids, ln
    print(kind, "  ".join(out), "ms per epoch; epochs_for_shape:", gtok.Graph2TrailTokenizer.epochs_for_shape(G, ld), flush=True)
