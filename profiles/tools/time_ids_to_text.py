"""Where gtok.ops.ids_to_text spends its time on the ZINC-full strings (249,456 rows, ~177 MB of text): host preparation of the
string table and suffixes, the length pass, the prefix sum, the byte pass."""
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
ids, ln = gtok.ops.ibtt_zinc(b, lut, 1 << 30, 0)
tail = [b" val_0_50 <eos>"] * G
for _ in range(2): blob, ptr = gtok.ops.ids_to_text(ids, ln, strings, tail)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); blob, ptr = gtok.ops.ids_to_text(ids, ln, strings, tail); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"{G} rows, ld {ids.shape[1]}, {blob.numel() / 1e6:.1f} MB of text: ids_to_text end to end {min(ts) * 1e3:.2f} ms (best of 5), {sorted(ts)[2] * 1e3:.2f} median")
y = torch.from_numpy(np.random.default_rng(3).normal(0, 2, G).astype(np.float32)).to(dev)
for _ in range(2): take, sb, sp = gtok.ops.zinc_text_tails(y, ln, 1 << 30); blob2, ptr2 = gtok.ops.ids_to_text(ids, take, strings, (sb, sp))
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); take, sb, sp = gtok.ops.zinc_text_tails(y, ln, 1 << 30); blob2, ptr2 = gtok.ops.ids_to_text(ids, take, strings, (sb, sp))
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"tails rendered on the device (gtok_zinc_text_tails, 2 launches + prefix sum) + ids_to_text: {min(ts) * 1e3:.2f} ms (best of 5), {sorted(ts)[2] * 1e3:.2f} median, "
      f"{blob2.numel() / 1e6:.1f} MB of text")
# the pieces
t0 = time.perf_counter()
enc = [t.encode("ascii") for t in strings]
tptr = np.zeros(len(enc) + 1, np.int32); np.cumsum([len(x) for x in enc], out=tptr[1:])
t1 = time.perf_counter()
sptr = np.zeros(G + 1, np.int64); np.cumsum(np.fromiter(map(len, tail), np.int64, G), out=sptr[1:])
sbytes = bytearray(b"".join(tail))
t2 = time.perf_counter()
sb = torch.frombuffer(sbytes, dtype=torch.uint8).to(dev); sp = torch.from_numpy(sptr).to(dev); torch.cuda.synchronize()
t3 = time.perf_counter()
print(f"host: table {1e3 * (t1 - t0):.2f} ms, suffix lengths + join {1e3 * (t2 - t1):.2f} ms, suffix H2D ({len(sbytes) / 1e6:.1f} + {sptr.nbytes / 1e6:.1f} MB) {1e3 * (t3 - t2):.2f} ms")
