"""sent_blane_kernel on one rank's share of config 5 (125 k ER graphs of 10-256 nodes, max_len 600): row flavours x K epochs."""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
d = gtok.synth.er_batch_device(G, dev, seed=1000)
b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], device=dev)
ld = 608
for K in (1, 2, 4, 8):
    for u16, pad in ((False, True), (True, False)):
        ids = torch.empty((K * G, ld), dtype=torch.int16 if u16 else torch.int32, device=dev)
        ln = torch.empty(K * G, dtype=torch.int32, device=dev)
        f = lambda k: gtok.ops.sent(b, b.max_nodes, 600, 0, k * K, ld=ld, out=(ids, ln), pad=pad, epochs=K, u16=u16)
        for _ in range(5): f(0)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for k in range(30 // K): f(k)
            e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / (30 // K))
        print(f"{gtok.ops.sent_kernel_name(b, b.max_nodes, 600, epochs=K)} K={K} {'u16' if u16 else 'i32'} {'padded' if pad else 'nopad '}: {best:.4f} ms per launch, {best / K:.4f} per epoch", flush=True)
        del ids, ln
