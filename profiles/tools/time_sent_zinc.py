"""ZINC-full sized SENT epoch under the lane kernel: batch reordered by walk length (default) vs as stored."""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
d = gtok.synth.zinc_like(249456, seed=1000)
host = gtok.GraphBatch.from_coo(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"])
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
ref = None
for nosort in ("1", "0"):
    os.environ["GTOK_NO_LANE_SORT"] = nosort
    b = host.to(dev)
    ids = torch.full((b.num_graphs, 208), -1, dtype=torch.int32, device=dev); ln = torch.empty(b.num_graphs, dtype=torch.int32, device=dev)
    for _ in range(3): gtok.ops.sent(b, 37, 1024, 0, 0, ld=208, out=(ids, ln), **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(200): gtok.ops.sent(b, 37, 1024, 0, k, ld=208, out=(ids, ln), **kw)
    e.record(); torch.cuda.synchronize()
    sb = b.lane_sorted
    print(f"sorted={nosort == '0'} units {sb.num_units if sb is not None else (b.num_graphs + 63) // 64} chunk {(sb or b).chunk_nodes}/{(sb or b).chunk_edges}: {s.elapsed_time(e)/200:.4f} ms", flush=True)
    if ref is None: ref = (ids.clone(), ln.clone())
    else: assert torch.equal(ref[0], ids) and torch.equal(ref[1], ln), "sorted and unsorted runs differ"
print("same tokens")
