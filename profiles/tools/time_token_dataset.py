"""Where TokenDataset.__init__ spends its time on the ZINC-full strings (249,456 examples): host loops, packing, the kernel,
the packed host copy, the row views."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
gdl = gtok.graph_data_loader
ops = gtok.ops
dev = torch.device("cuda", 0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 249456
d = gtok.synth.zinc_like(G, seed=1000)
pyg = gtok.synth.InMemoryLike(d)
zds = gdl.ZINCTokenizationDataset(split="train", max_len=1024, zinc_dataset=pyg)
ex = [zds[i] for i in range(G)]
vocab = gdl.build_zinc_vocab_on_device([e["text"] for e in ex], device=dev)
def clock(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return time.perf_counter() - t0, r
t_all, td = clock(lambda: gdl.TokenDataset(ex, vocab, 1024, device=dev))
t_all2, td = clock(lambda: gdl.TokenDataset(ex, vocab, 1024, device=dev))
t_loop, (texts, labels) = clock(lambda: ([e["text"] for e in ex], [int(e["label"]) for e in ex]))
t_y, _ = clock(lambda: torch.tensor(labels, dtype=torch.long))
t_pack, (blob, ptr) = clock(lambda: ops.pack_texts(texts))
t_h2d, blob_d = clock(lambda: blob.to(dev))
t_tab, table = clock(lambda: ops.VocabTable(vocab, dev))
t_k, (ids, lens) = clock(lambda: ops.text_to_ids(blob_d, ptr, table, 1024, True))
t_rows, rows = clock(lambda: gtok.rows.EpochRows(ids, lens, pin=False))
t_views, seqs = clock(lambda: [td.seqs[i] for i in range(len(td))])
print(f"TokenDataset.__init__: {t_all:.3f} s first, {t_all2:.3f} s again = example loop {t_loop:.3f} + label tensors {t_y:.3f} + pack_texts {t_pack:.3f} "
      f"+ H2D {t_h2d:.3f} + vocab table {t_tab:.3f} + text_to_ids {t_k:.4f} + packed host copy {t_rows:.3f}; every row view on demand afterwards: {t_views:.3f}")
