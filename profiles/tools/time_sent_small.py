"""Small-batch labelled SENT (BASELINE config 2: 12 k molecules; config 4's per-rank share on 8 GPUs: 31 k): kernel
chosen by the launcher and each pinned kernel, HIP events around 50 launches."""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
for G in (12000, 31182):
    d = gtok.synth.zinc_like(G, seed=1000)
    ref = None
    for pin in ("", "reg", "lane"):
        if pin: os.environ["GTOK_SENT_KERNEL"] = pin
        else: os.environ.pop("GTOK_SENT_KERNEL", None)
        b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)
        ids = torch.empty((G, 208), dtype=torch.int32, device=dev); ln = torch.empty(G, dtype=torch.int32, device=dev)
        for _ in range(5): gtok.ops.sent(b, 37, 1024, 0, 0, ld=208, out=(ids, ln), **kw)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for k in range(50): gtok.ops.sent(b, 37, 1024, 0, k, ld=208, out=(ids, ln), **kw)
        e.record(); torch.cuda.synchronize()
        name = gtok.ops.sent_kernel_name(b, 37, 1024, **kw)
        print(f"G={G} pin={pin or '-':5s} {name:18s} {s.elapsed_time(e) / 50:.4f} ms", flush=True)
        if ref is None: ref = (ids.clone(), ln.clone())
        else: assert torch.equal(ref[0], ids) and torch.equal(ref[1], ln), "kernels differ"
os.environ.pop("GTOK_SENT_KERNEL", None)
