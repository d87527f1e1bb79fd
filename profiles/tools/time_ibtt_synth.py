"""gtok_ibtt_synth on the config-5 share (125 k ER graphs of 10-256 nodes, max_len 600): best of 3 x 50 back-to-back launches."""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
d = gtok.synth.er_batch_device(G, dev, seed=1000)
b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], device=dev)
vocab = {t: i for i, t in enumerate(["<pad>", "<bos>", "<e>", "<n>", "<q>", "<p>", "<eos>", "yes", "no"] + [str(i) for i in range(b.max_nodes)])}
lut = gtok.ops.synth_lut(vocab, b.max_nodes).to(dev)
ids = torch.empty((G, 600), dtype=torch.int32, device=dev); ln = torch.empty(G, dtype=torch.int32, device=dev)
for _ in range(5): gtok.ops.ibtt_synth(b, lut, None, 600, vocab["<pad>"], ld=600, out=(ids, ln))
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(50): gtok.ops.ibtt_synth(b, lut, None, 600, vocab["<pad>"], ld=600, out=(ids, ln))
    e.record(); torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) / 50)
print(f"{os.path.basename(gtok._lib.LIB_PATH)}: {best:.4f} ms  (checksum {int(ids.sum())}, tokens {int(ln.sum())})", flush=True)
