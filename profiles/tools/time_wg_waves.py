"""sent_lane_kernel per-epoch times over the launch shapes the round cares about, for one setting of GTOK_LANE_WG_WAVES (printed):
run once per setting in the same GPU call."""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
def corpus(G, seed=1000):
    d = gtok.synth.zinc_like(G, seed=seed)
    return gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)
def run(b, G, K, u16, pad, n):
    ids = torch.empty((K * G, 208), dtype=torch.int16 if u16 else torch.int32, device=dev); ln = torch.empty(K * G, dtype=torch.int32, device=dev)
    for _ in range(max(3, n // 10)): gtok.ops.sent(b, 37, 1024, 0, 0, ld=208, out=(ids, ln), epochs=K, u16=u16, pad=pad, **kw)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for k in range(n): gtok.ops.sent(b, 37, 1024, 0, k * K, ld=208, out=(ids, ln), epochs=K, u16=u16, pad=pad, **kw)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n / K)
    return best
out = []
for G, shapes in ((249456, ((1, False, True), (1, False, False), (1, True, True), (1, True, False), (2, True, False), (4, True, False))),
                  (124728, ((1, False, True), (2, True, False), (8, True, False))),
                  (62364, ((1, False, True), (2, True, False), (4, True, False), (16, True, False))),
                  (31182, ((4, True, False), (8, False, True), (8, True, False), (32, True, False))),
                  (12000, ((16, True, False), (21, True, False), (24, False, True), (32, True, False)))):
    b = corpus(G)
    for K, u16, pad in shapes:
        out.append(f"{G}x{K} {'u16' if u16 else 'i32'}{'p' if pad else 'n'} {run(b, G, K, u16, pad, 100 if G * K < 600000 else 40) * 1e3:.2f}us")
    del b
print("WG_WAVES=" + os.environ.get("GTOK_LANE_WG_WAVES", "default"), " | ".join(out), flush=True)
