"""sent_blane_kernel on the config-5 shares (125 k graphs of 10-256 nodes, max_len 600): one epoch per launch as the padded int32 slab
(the bench headline of synth_er / synth_mix) and K = 14 epochs per launch as 16-bit rows (what the dataset classes run).
python profiles/tools/time_blane_r05.py [er|mix ...]"""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
G, ld = 125000, 608
for wl in (sys.argv[1:] or ["er", "mix"]):
    d = (gtok.synth.mix_batch_device if wl == "mix" else gtok.synth.er_batch_device)(G, dev, seed=1000)
    b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], device=dev)
    out = []
    for K, u16, pad in ((1, False, True), (1, True, False), (14, True, False)):
        ids = torch.empty((K * G, ld), dtype=torch.int16 if u16 else torch.int32, device=dev); ln = torch.empty(K * G, dtype=torch.int32, device=dev)
        f = lambda k: gtok.ops.sent(b, 256, 600, 0, k * K, ld=ld, out=(ids, ln), epochs=K, u16=u16, pad=pad)
        for _ in range(2): f(0)
        torch.cuda.synchronize(); best = 1e9
        n = 20 if K == 1 else 4
        for rep in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for k in range(n): f(k)
            e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e) / n / K)
        out.append(f"K={K} {'u16 nopad' if u16 else 'int32 padded'}: {best:.4f} ms/epoch")
        del ids, ln
    print(f"{wl}: {gtok.ops.last_sent_kernel()}  " + "  ".join(out), flush=True)
