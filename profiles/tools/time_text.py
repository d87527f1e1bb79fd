"""Time gtok_text_to_ids at scale: the graph-token texts of `base` ER graphs (10-256 nodes), tiled `rep` times."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
base, rep = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 16
d = gtok.synth.er_batch_device(base, dev, seed=1000)
eptr = np.concatenate([[0], np.cumsum(d["edge_counts"])])
texts = []
for g in range(base):
    u = d["src"][eptr[g]:eptr[g + 1]].tolist(); v = d["dst"][eptr[g]:eptr[g + 1]].tolist()
    body = " ".join(f"{a} {b} <e>" for a, b in zip(u, v))
    texts.append(" ".join(t for t in ("<bos>", body, "<n>", " ".join(map(str, range(int(d["node_counts"][g])))), "<q> has_cycle <p> yes <eos>") if t))
vocab = {t: i for i, t in enumerate(["<pad>", "<bos>", "<e>", "<n>", "<q>", "<p>", "<eos>", "yes", "no", "has_cycle"] + [str(i) for i in range(256)])}
tb, tp = gtok.ops.pack_texts(texts)
L = int(tb.numel())
tb = tb.repeat(rep).to(dev)
tp = torch.cat([tp[:-1] + k * L for k in range(rep)] + [torch.tensor([rep * L])]).to(dev)
table = gtok.ops.VocabTable(vocab, dev)
G = base * rep
ids = torch.empty((G, 600), dtype=torch.int32, device=dev); ln = torch.empty(G, dtype=torch.int32, device=dev)
need = sum(len(" ".join(t.split()[:600])) for t in texts) * rep
for _ in range(3):
    gtok.ops.text_to_ids(tb, tp, table, 600, True, ld=600, out=(ids, ln))
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    gtok.ops.text_to_ids(tb, tp, table, 600, True, ld=600, out=(ids, ln))
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print(f"{G} texts, {tb.numel() / 1e6:.0f} MB of text ({need / 1e6:.0f} MB up to the cut): {ms:.4f} ms  {G / ms / 1e3:.1f} M texts/s  "
      f"{(need + 4.0 * float(ln.sum()) ) / ms / 1e6:.0f} GB/s algorithmic")

# the same texts through the device parser (text -> edges / nodes / query / label: the whole text is read)
r = gtok.ops.parse_graph_texts(tb, tp); torch.cuda.synchronize()
assert int(r["status"].sum()) == 0
s2, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s2.record()
for _ in range(3):
    r = gtok.ops.parse_graph_texts(tb, tp)
e2.record(); torch.cuda.synchronize()
ms2 = s2.elapsed_time(e2) / 3
print(f"parse_graph_texts (two passes + prefix sum): {ms2:.3f} ms  {G / ms2 / 1e3:.1f} M texts/s  {tb.numel() * 2 / ms2 / 1e6:.0f} GB/s of text read, "
      f"{int(r['edge_ptr'][-1])} edges")

if os.environ.get("GTOK_PHASE"):
    ids2 = torch.empty((G, 608), dtype=torch.int32, device=dev)
    gtok.ops.text_to_ids(tb, tp, table, 600, True, ld=608, out=(ids2, ln)); torch.cuda.synchronize()
    ph = ids2[:, -4:].double()
    npieces = ph[:, 3].sum()
    print("per piece: classify", float(ph[:, 0].sum() / npieces), "scan", float(ph[:, 1].sum() / npieces), "probe", float(ph[:, 2].sum() / npieces), "pieces/text", float(ph[:, 3].mean()))
