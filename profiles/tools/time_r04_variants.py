"""A/B of sent_lane_kernel build variants (GTOK_LIB) on the regimes round 4 cares about: ZINC-full (one round of resident waves),
1 M molecules and 31 k x 21 epochs (several rounds)."""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)


def corpus(G, seed=1000):
    d = gtok.synth.zinc_like(G, seed=seed)
    return gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)


def timeit(f, n, reps=3):
    for _ in range(max(3, n // 10)): f(0)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for k in range(n): f(k)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n)
    return best


def run(b, G, ld, K=1, u16=False, pad=True, n=200):
    ids = torch.empty((K * G, ld), dtype=torch.int16 if u16 else torch.int32, device=dev)
    ln = torch.empty(K * G, dtype=torch.int32, device=dev)
    return timeit(lambda k: gtok.ops.sent(b, 37, 1024, 0, k * K, ld=ld, out=(ids, ln), pad=pad, epochs=K, u16=u16, **kw), n)


tag = os.path.basename(gtok._lib.LIB_PATH)
out = []
b = corpus(249456)
for u16, pad in ((False, True), (False, False), (True, True), (True, False)):
    out.append(f"zinc {'u16' if u16 else 'i32'}{'p' if pad else 'n'} {run(b, 249456, 208, u16=u16, pad=pad):.4f}")
del b
b = corpus(31182)
for u16, pad in ((False, True), (True, False)):
    out.append(f"31kx21 {'u16' if u16 else 'i32'}{'p' if pad else 'n'} {run(b, 31182, 208, K=21, u16=u16, pad=pad, n=30):.4f}")
del b
b = corpus(1000000, seed=3)
for u16, pad in ((False, True), (False, False), (True, True), (True, False)):
    out.append(f"1M {'u16' if u16 else 'i32'}{'p' if pad else 'n'} {run(b, 1000000, 200, u16=u16, pad=pad, n=30):.4f}")
print(tag, " | ".join(out), flush=True)
