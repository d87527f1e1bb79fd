"""gtok_pack_rows_scan / gtok_unpack_rows_u16 on config-5-shaped rows (125 k ER graphs of 10-256 nodes, rows cut at 600 tokens, K epochs of
16-bit rows from sent_blane_kernel): HIP events, median of 10.  python profiles/tools/time_pack_er.py [K ...]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda:0")
G, ld = 125000, 608
d = gtok.synth.er_batch_device(G, dev, seed=1000)
b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], device=dev)


def ev(f, n=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


for K in [int(a) for a in sys.argv[1:]] or [1, 14]:
    ids = torch.empty((K * G, ld), dtype=torch.int16, device=dev); ln = torch.empty(K * G, dtype=torch.int32, device=dev)
    walk = lambda: gtok.ops.sent(b, 256, 600, 0, 0, ld=ld, out=(ids, ln), pad=False, epochs=K, u16=True)
    walk(); torch.cuda.synchronize()
    toks = int(ln.clamp(0, ld).sum())
    cap = K * G * ld
    t_walk = ev(walk)
    t_scan = ev(lambda: gtok.ops.pack_rows_scan(ids, ln, 2, cap))
    packed, ptr, st = gtok.ops.pack_rows_scan(ids, ln, 2, cap)
    out = torch.empty_like(ids)
    t_un = ev(lambda: gtok.ops.unpack_rows(packed, ptr, ln, ld, 5, out=out, u16=True))
    print(f"K={K:2d} ({gtok.ops.last_sent_kernel()}): walk {t_walk:.4f} ms ({t_walk / K:.4f} per epoch) | pack_rows_scan {t_scan:.4f} ({t_scan / K:.4f} per epoch, "
          f"{4 * toks / t_scan / 1e9:.2f} TB/s of in + out) | unpack into a 16-bit slab {t_un:.4f} ({(2 * toks + out.numel() * 2) / t_un / 1e9:.2f} TB/s)", flush=True)
    del ids, ln, packed, out
