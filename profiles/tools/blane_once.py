"""A few launches of sent_blane_kernel on the synth_er share (counter passes: rocprofv3 --pmc ... -- python3 profiles/tools/blane_once.py [K])."""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
G, ld = 125000, 608
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
wl = sys.argv[2] if len(sys.argv) > 2 else "er"
d = (gtok.synth.mix_batch_device if wl == "mix" else gtok.synth.er_batch_device)(G, dev, seed=1000)
b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], device=dev)
ids = torch.empty((K * G, ld), dtype=torch.int16, device=dev); ln = torch.empty(K * G, dtype=torch.int32, device=dev)
for k in range(3):
    gtok.ops.sent(b, 256, 600, 0, k * K, ld=ld, out=(ids, ln), epochs=K, u16=True, pad=False)
torch.cuda.synchronize()
print(gtok.ops.last_sent_kernel(), "tokens", int(ln.sum()))
