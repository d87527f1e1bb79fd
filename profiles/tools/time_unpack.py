"""The re-padding pass of the compact exchange (gtok_unpack_rows_at / gtok_unpack_rows_u16) on K epochs of ZINC-full rows packed by
the walk: HIP events, median of 20.  python profiles/tools/time_unpack.py [K ...]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda:0")
G, ld = 249456, 176
d = gtok.synth.zinc_like(G, seed=1000)
b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)


def ev(f, n=20):
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
    pk = gtok.ops.PackedRows(K * G, K * G * 96, True, dev)
    _, ln = gtok.ops.sent(b, 37, 1024, 0, 0, ld=ld, epochs=K, u16=True, packed=pk, slab=False, **kw)
    ln = ln.reshape(-1)
    out = torch.empty((K * G, ld), dtype=torch.int16, device=dev)
    out32 = torch.empty((K * G, ld), dtype=torch.int32, device=dev) if K <= 4 else None
    used = int(pk.used())
    t = ev(lambda: gtok.ops.unpack_rows_at(pk.buf, pk.row_start, ln, ld, 5, out=out, u16=True))
    mb = (used * 2 + out.numel() * 2) / 1e6
    line = f"K={K:2d}: unpack_rows_at -> 16-bit slab {t:.4f} ms ({t / K * 1e3:.1f} us per epoch, {mb / t / 1e3:.2f} TB/s of {mb:.0f} MB in + out)"
    if out32 is not None:
        t32 = ev(lambda: gtok.ops.unpack_rows_at(pk.buf, pk.row_start, ln, ld, 5, out=out32))
        mb32 = (used * 2 + out32.numel() * 4) / 1e6
        line += f"; -> int32 slab {t32:.4f} ms ({mb32 / t32 / 1e3:.2f} TB/s of {mb32:.0f} MB)"
    print(line, flush=True)
    del pk, out, out32
