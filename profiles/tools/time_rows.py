"""Packed rows on a ZINC-full epoch: row_offsets / pack (16 and 32 bit) / unpack timed with HIP events, bytes moved,
and the D2H copy of the packed form against the D2H copy of the padded slab (what TokenizedGraphDataset serves from)."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 249456
d = gtok.synth.zinc_like(G, seed=1000)
b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
ids, ln = gtok.ops.sent(b, 37, 1024, 0, 0, ld=208, **kw)
torch.cuda.synchronize()


def timed(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


ptr = gtok.ops.row_offsets(ln, 208)
total = int(ptr[-1]); toks = int(ln.sum())
print(f"{G} rows, {toks} tokens, packed {total} elements ({total * 2 / 1e6:.1f} MB at 16 bits; slab {ids.numel() * 4 / 1e6:.1f} MB)")
print(f"row_offsets      {timed(lambda: gtok.ops.row_offsets(ln, 208)):.4f} ms")
for eb in (2, 4):
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    packed = torch.empty(total, dtype=torch.int16 if eb == 2 else torch.int32, device=dev)
    f = lambda: gtok.lib().gtok_pack_rows(ids.data_ptr(), 208, ln.data_ptr(), G, ptr.data_ptr(), eb, packed.data_ptr(), total, st.data_ptr(), None)
    t = timed(f)
    print(f"pack_rows {eb}B     {t:.4f} ms  ({(toks * 4 + total * eb) / t / 1e6:.0f} GB/s read+write)")
    out = torch.empty_like(ids)
    t = timed(lambda: gtok.ops.unpack_rows(packed, ptr, ln, 208, 5, out=out))
    print(f"unpack_rows {eb}B   {t:.4f} ms  ({(total * eb + ids.numel() * 4) / t / 1e6:.0f} GB/s read+write)")
    assert torch.equal(torch.where(torch.arange(208, device=dev)[None, :] < ln[:, None], ids, 5), out)
    pin = torch.empty(packed.shape, dtype=packed.dtype, pin_memory=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): pin.copy_(packed, non_blocking=True); torch.cuda.synchronize()
    print(f"D2H packed {eb}B    {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
pin = torch.empty(ids.shape, dtype=ids.dtype, pin_memory=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): pin.copy_(ids, non_blocking=True); torch.cuda.synchronize()
print(f"D2H padded slab  {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms (pinned); pageable .cpu(): ", end="")
t0 = time.perf_counter(); ids.cpu(); print(f"{(time.perf_counter() - t0) * 1e3:.1f} ms")
