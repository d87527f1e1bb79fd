"""One-off scale check: 1,000,000 ZINC-shaped molecules in one launch, SENT and IBTT bit-exact against the oracle."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import both, gtok, orc, zinc_vocab
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
t0 = time.time()
d = gtok.synth.zinc_like(G, seed=3)
batch, coo = both(d)
dev = batch.to("cuda:0")
print(f"corpus {G} graphs, {batch.num_nodes_total} nodes, {batch.num_edges_total} entries, built in {time.time() - t0:.1f}s", flush=True)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
T = min(64, orc.num_threads())
ids, ln = gtok.ops.sent(dev, 37, 1024, 5, 1, ld=200, **kw)
ref, rln = orc.sent(coo, 37, 1024, 5, 1, ld=200, nthreads=T, **kw)
assert int(ln.max()) <= 200 and np.array_equal(ln.cpu().numpy(), rln) and np.array_equal(ids.cpu().numpy(), ref)
print("SENT bit-exact", gtok.ops.sent_kernel_name(dev, 37, 1024, **kw), flush=True)
vocab = zinc_vocab(40); lut = gtok.ops.zinc_lut(vocab, 40)
ids, ln = gtok.ops.ibtt_zinc(dev, lut, 1024, vocab["<pad>"], ld=240)
ref, rln = orc.ibtt_zinc(coo, lut.numpy(), 1024, vocab["<pad>"], 240, nthreads=T)
assert np.array_equal(ln.cpu().numpy(), rln) and np.array_equal(ids.cpu().numpy(), ref)
print("IBTT bit-exact", flush=True)
# rows packed by the walk (gtok_sent_packed, no slab): four rounds of resident waves reuse the staging rows; re-padded, the rows are the oracle's
need = int(((torch.from_numpy(rln_sent := orc.sent(coo, 37, 1024, 5, 1, ld=200, nthreads=T, **kw)[1]).clamp(0, 200) + 7) // 8 * 8).sum())
pk = gtok.ops.PackedRows(G, int(need * 1.04) + 4096, True, "cuda:0")
_, pln = gtok.ops.sent(dev, 37, 1024, 5, 1, ld=200, u16=True, packed=pk, slab=False, **kw)
back = gtok.ops.unpack_rows_at(pk.buf, pk.row_start, pln, 200, 5, u16=True).cpu().numpy().view(np.uint16).astype(np.int32)
assert pk.fused and int(pk.status()) == 0 and np.array_equal(pln.cpu().numpy(), rln_sent) and np.array_equal(back, orc.sent(coo, 37, 1024, 5, 1, ld=200, nthreads=T, **kw)[0])
print("SENT packed by the walk bit-exact (no slab)", flush=True)
del back
flavours = [("sent 16-bit rows packed by the walk, no slab (gtok_sent_packed + GTOK_SENT_PACK_ONLY)",
             lambda: gtok.ops.sent(dev, 37, 1024, 5, 1, ld=200, u16=True, packed=pk, slab=False, **kw)),
            ("sent int32 padded (the documented slab)", lambda: gtok.ops.sent(dev, 37, 1024, 5, 1, ld=200, **kw)),
            ("sent int32 unpadded", lambda: gtok.ops.sent(dev, 37, 1024, 5, 1, ld=200, pad=False, **kw)),
            ("sent 16-bit rows padded", lambda: gtok.ops.sent(dev, 37, 1024, 5, 1, ld=200, u16=True, **kw)),
            ("sent 16-bit rows unpadded", lambda: gtok.ops.sent(dev, 37, 1024, 5, 1, ld=200, u16=True, pad=False, **kw)),
            ("ibtt", lambda: gtok.ops.ibtt_zinc(dev, lut, 1024, vocab["<pad>"], ld=240))]
for name, f in flavours:
    for _ in range(3): f()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): f()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 20)
    print(f"{name}: {best:.4f} ms  {G / best / 1e3:.0f} M graphs/s", flush=True)
