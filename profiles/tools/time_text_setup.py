"""gtok_text_to_ids: what a workgroup pays before its first text (the LDS tables are built per workgroup).  A launch over a few
texts is all set-up: small vocab (table in LDS) vs the ZINC trainer's vocab with thousands of label tokens (capacity 8192)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
words = ["<bos>", "<eos>", "<pad>", "<unk>", "<q>", "<p>", "<atom>", "<bond>", "C", "N", "O", "single", "aromatic", "regression"] + [str(i) for i in range(40)]
small = {t: i for i, t in enumerate(words)}
big = dict(small)
for k in range(3000): big.setdefault(f"val_{k}_{k % 100:02d}", len(big))
def ev(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for G in (64, 2048, 16384):
    texts = [("<bos> " + " ".join(words[(i + j) % len(words)] for j in range(100)) + " <p> val_1_01 <eos>")for i in range(G)]
    blob, ptr = gtok.ops.pack_texts(texts); blob, ptr = blob.to(dev), ptr.to(dev)
    out = (torch.empty((G, 128), dtype=torch.int32, device=dev), torch.empty(G, dtype=torch.int32, device=dev))
    for name, v in (("small", small), ("big", big)):
        tb = gtok.ops.VocabTable(v, dev)
        t = ev(lambda: gtok.ops.text_to_ids(blob, ptr, tb, 1024, True, ld=128, out=out))
        print(f"{G:6d} texts, {name:5s} vocab (capacity {tb.capacity}): {t * 1e3:.1f} us")
