"""gtok_text_to_ids on the rendered strings of the ZINC-full corpus (what TokenDataset.__init__ launches in the zero-edit IBTT
flow): 249,456 texts of ~600 bytes.  Also times the renderer's two launches (gtok_ibtt_zinc with the identity LUT +
gtok_ids_to_text)."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 249456
d = gtok.synth.zinc_like(G, seed=1000)
b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)
strings = ["<bos>", "<eos>", "<atom>", "<bond>", "<q>", "regression", "<p>"] + list(gtok.ops.ZINC_ATOM_SYMBOLS) + list(gtok.ops.ZINC_BOND_NAMES) + [str(i) for i in range(b.max_nodes)]
lut = torch.arange(len(strings), dtype=torch.int32)
def ev(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): r = f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n, r
t_ids, (ids, ln) = ev(lambda: gtok.ops.ibtt_zinc(b, lut, 1 << 30, 0))
tail = [b" val_0_50 <eos>"] * G
t0 = time.perf_counter(); blob, ptr = gtok.ops.ids_to_text(ids, ln, strings, tail); torch.cuda.synchronize(); t_txt = time.perf_counter() - t0
vocab = {t: i for i, t in enumerate(["<bos>", "<eos>", "<pad>", "<unk>", "<q>", "<p>", "<atom>", "<bond>", "C", "N", "O", "F", "P", "S", "Cl", "Br", "I",
                                      "single", "double", "triple", "aromatic", "regression"] + [str(i) for i in range(b.max_nodes)] + ["X", "unknown"])}
table = gtok.ops.VocabTable(vocab, dev)
out = (torch.empty((G, 240), dtype=torch.int32, device=dev), torch.empty(G, dtype=torch.int32, device=dev))
t_tok, _ = ev(lambda: gtok.ops.text_to_ids(blob, ptr, table, 1024, True, ld=240, out=out))
nbytes = int(blob.numel()); ntok = int(out[1].sum())
print(f"{G} texts, {nbytes / 1e6:.0f} MB: ibtt_zinc(identity LUT) {t_ids:.3f} ms; ids_to_text (2 launches + host prep) {t_txt * 1e3:.1f} ms; "
      f"text_to_ids {t_tok:.3f} ms = {(nbytes + 4 * ntok) / t_tok / 1e6:.0f} GB/s ({(nbytes + 4 * ntok) / t_tok / 1e6 / 8000:.3f} of 8 TB/s), {ntok / G:.1f} tokens per text")

# the vocab the ZINC trainer really builds: thousands of label tokens on top (capacity > 1024: the full table stays in global memory)
big = dict(vocab)
for k in range(3000): big.setdefault(f"val_{k}_{k % 100:02d}", len(big))
tb = gtok.ops.VocabTable(big, dev)
t_big, _ = ev(lambda: gtok.ops.text_to_ids(blob, ptr, tb, 1024, True, ld=240, out=out))
print(f"same texts, vocab of {len(big)} keys (capacity {tb.capacity}): text_to_ids {t_big:.3f} ms = {(nbytes + 4 * ntok) / t_big / 1e6:.0f} GB/s")
