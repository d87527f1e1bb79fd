"""Round 4 timings of gtok_sent (back-to-back launches, best of 3 x N): the ZINC-full corpus as int32 / 16-bit rows, padded or
not; K epochs per launch of the 12 k (BASELINE config 2) and 31 k (one rank's share of config 4 on 8 GPUs) splits; 1 M
molecules (more than one round of resident waves).  Usage: time_r04.py [zinc] [epochs] [1m]"""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
what = set(sys.argv[1:]) or {"zinc", "epochs", "1m"}
v4 = gtok.lib().gtok_version() >= 4


def corpus(G, seed=1000):
    d = gtok.synth.zinc_like(G, seed=seed)
    return gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)


def timeit(f, n, reps=3):
    for _ in range(max(3, n // 10)): f(0)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for k in range(n): f(k)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n)
    return best


def run(b, G, ld, K=1, u16=False, pad=True, n=200):
    ids = torch.empty((K * G, ld), dtype=torch.int16 if u16 else torch.int32, device=dev)
    ln = torch.empty(K * G, dtype=torch.int32, device=dev)
    extra = dict(epochs=K, u16=u16) if v4 else {}
    return timeit(lambda k: gtok.ops.sent(b, 37, 1024, 0, k * K, ld=ld, out=(ids, ln), pad=pad, **extra, **kw), n)


print(os.path.basename(gtok._lib.LIB_PATH), "ABI", gtok.lib().gtok_version(), flush=True)
if "zinc" in what:
    G = 249456
    b = corpus(G)
    for u16 in ((False, True) if v4 else (False,)):
        for pad in (True, False):
            ms = run(b, G, 208, u16=u16, pad=pad)
            print(f"zinc_full {G}: {'u16' if u16 else 'i32'} {'padded' if pad else 'nopad '}: {ms:.4f} ms  {G / ms / 1e6:.2f} G graphs/s", flush=True)
    del b
if "epochs" in what and v4:
    for G in (12000, 31182):
        b = corpus(G)
        for K in (1, 4, 8, 16, 21, 24, 32, 48, 64):
            if K * G > 2100000: continue
            name = gtok.ops.sent_kernel_name(b, 37, 1024, epochs=K, **kw)
            for u16, pad in ((False, True), (True, False)):
                ms = run(b, G, 208, K=K, u16=u16, pad=pad, n=max(20, 400 // K))
                print(f"{G} x K={K:2d} {name:16s} {'u16 nopad' if u16 else 'i32 pad  '}: {ms:.4f} ms per launch, {ms / K:.5f} ms per epoch, {G * K / ms / 1e6:.2f} G graphs/s", flush=True)
        del b
if "1m" in what:
    G = 1000000
    b = corpus(G, seed=3)
    for u16, pad in (((False, True), (False, False), (True, True), (True, False)) if v4 else ((False, True), (False, False))):
        ms = run(b, G, 200, u16=u16, pad=pad, n=30)
        print(f"1M molecules: {'u16' if u16 else 'i32'} {'padded' if pad else 'nopad '}: {ms:.4f} ms  {G / ms / 1e6:.2f} G graphs/s", flush=True)
