"""The packed form of a 16-bit SENT slab, pass by pass (HIP events, median of 30): gtok_row_offsets, gtok_pack_rows_u16 and the
one-pass gtok_pack_rows_scan, on ZINC-full-shaped rows of K epochs.  python profiles/tools/time_pack.py [K ...]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda:0")
G, ld = 249456, 176
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
    walk(); torch.cuda.synchronize()
    cap = K * G * ld
    ptr = gtok.ops.row_offsets(ln, ld)
    tokens = int(ln.clamp(0, ld).sum())
    t_walk = ev(walk)
    t_off = ev(lambda: gtok.ops.row_offsets(ln, ld))
    t_pack = ev(lambda: gtok.ops.pack_rows_u16(ids, ln, ptr, elem_bytes=2, capacity=cap, check_status=False))
    t_scan = ev(lambda: gtok.ops.pack_rows_scan(ids, ln, 2, cap))
    t_both = ev(lambda: (walk(), gtok.ops.pack_rows_scan(ids, ln, 2, cap)))
    pk = gtok.ops.PackedRows(K * G, K * G * 96, True, dev)
    fused = lambda: gtok.ops.sent(b, b.max_nodes, 1024, 0, 0, ld=ld, out=(ids, ln), pad=False, epochs=K, u16=True, packed=pk, **kw)
    t_fused = ev(fused)
    assert pk.fused and int(pk.status()) == 0
    mb = 2 * tokens * 2 / 1e6
    print(f"K={K:2d} rows {K * G} tokens {tokens}: walk {t_walk:.4f} ms | row_offsets {t_off:.4f} | pack_rows_u16 {t_pack:.4f} | pack_rows_scan {t_scan:.4f} "
          f"({mb / t_scan / 1e3:.2f} TB/s of {mb:.0f} MB in+out) | walk + scan {t_both:.4f} | walk that packs (gtok_sent_packed, + zeroing its state) {t_fused:.4f}; per epoch: walk {t_walk / K:.5f} scan {t_scan / K:.5f} "
          f"fused {t_fused / K:.5f} (+{(t_fused - t_walk) / K * 1e3:.2f} us over the walk)")
    del ids, ln
