"""Time gtok_ibtt_zinc on the ZINC-shaped corpus at several slab widths (a narrow slab drops most stores:
separates store cost from the serialiser's instruction/latency cost)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 249456
d = gtok.synth.zinc_like(G, seed=1000)
host = gtok.GraphBatch.from_coo(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"])
b = host.to(dev)
max_nodes = host.max_nodes
vocab = {t: i for i, t in enumerate(
    ["<bos>", "<eos>", "<pad>", "<unk>", "<q>", "<p>", "<atom>", "<bond>", "C", "N", "O", "F", "P", "S", "Cl",
     "Br", "I", "single", "double", "triple", "aromatic", "regression"]
    + [str(i) for i in range(max_nodes)] + ["X", "unknown"])}
lut = gtok.ops.zinc_lut(vocab, max_nodes).to(dev)
for ld in (240, 64, 8):
    ids = torch.empty((G, ld), dtype=torch.int32, device=dev); ln = torch.empty(G, dtype=torch.int32, device=dev)
    for _ in range(3):
        gtok.ops.ibtt_zinc(b, lut, 1024, vocab["<pad>"], ld=ld, out=(ids, ln))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(20):
        gtok.ops.ibtt_zinc(b, lut, 1024, vocab["<pad>"], ld=ld, out=(ids, ln))
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f"{os.environ.get('GTOK_IBTT_KERNEL', 'auto'):5s} ld {ld:4d}  {ms:7.4f} ms  {G / ms / 1e3:8.1f} M graphs/s  max len {int(ln.max())}", flush=True)
