"""ZINC-full (249,456 molecules) x K epochs per launch x row flavours: does a launch of more than one round of resident waves
(staging and padding of some units overlapped with the walks of others) beat K launches of one round?"""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
G = 249456
d = gtok.synth.zinc_like(G, seed=1000)
b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)
for K in (1, 2, 3, 4):
    for u16, pad in ((False, True), (False, False), (True, True), (True, False)):
        ids = torch.empty((K * G, 208), dtype=torch.int16 if u16 else torch.int32, device=dev)
        ln = torch.empty(K * G, dtype=torch.int32, device=dev)
        f = lambda k: gtok.ops.sent(b, 37, 1024, 0, k * K, ld=208, out=(ids, ln), pad=pad, epochs=K, u16=u16, **kw)
        for _ in range(10): f(0)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for k in range(100 // K): f(k)
            e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / (100 // K))
        print(f"K={K} {'u16' if u16 else 'i32'} {'padded' if pad else 'nopad '}: {best:.4f} ms per launch, {best / K:.4f} ms per epoch, {G * K / best / 1e6:.2f} G graphs/s", flush=True)
        del ids, ln
