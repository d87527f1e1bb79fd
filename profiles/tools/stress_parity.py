"""One-off randomized parity sweep: many seeds / sizes / slab widths, every SENT and IBTT kernel pin, against the oracle."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import both, gtok, orc, zinc_vocab
DEV = "cuda:0"
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
T = min(32, orc.num_threads())
fails = 0
def cmp(tag, ids, ln, ref, rln):
    global fails
    if not (np.array_equal(ln.cpu().numpy(), rln) and np.array_equal(ids.cpu().numpy(), ref)):
        fails += 1; print("MISMATCH", tag, flush=True)
t0 = time.time()
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    seed = int(rng.integers(0, 2 ** 31))
    kind = it % 4
    if kind == 0:
        d = gtok.synth.zinc_like(int(rng.integers(1, 9000)), seed=seed, coalesced=bool(rng.integers(0, 2))); labeled = True; nn = 37
    elif kind == 1:
        d = gtok.synth.zinc_like(int(rng.integers(65536, 90000)), seed=seed); labeled = bool(rng.integers(0, 2)); nn = 40
    elif kind == 2:
        mx = int(rng.integers(12, 300))
        d = gtok.synth.er_batch(int(rng.integers(1, 700)), seed=seed, min_nodes=int(rng.integers(1, 11)), max_nodes=mx,
                                min_sparsity=0.02, max_sparsity=float(rng.uniform(0.05, 0.3))); labeled = False; nn = mx
    else:
        d = gtok.synth.graph_token_like(int(rng.integers(1, 300)), seed=seed, with_text=False); labeled = False; nn = 49
    batch, coo = both(d, labeled)
    dev = batch.to(DEV)
    max_len = int(rng.choice([1, 2, 7, 33, 100, 600, 1024, 4096]))
    kw = dict(labeled=labeled, num_node_types=28 if labeled else 0, num_edge_types=5 if labeled else 0,
              remap_zinc=bool(labeled and rng.integers(0, 2)))
    sd, ep, base = int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 1000)), int(rng.integers(0, 2 ** 40))
    pins = ["", "reg", "lds"] if batch.max_nodes <= 64 else ["", "lds"]
    if (batch.flags & 1) and batch.max_nodes <= 64 and batch.max_edges <= 255:
        pins += ["lane", "lane-unsorted", "lane-int32"]   # reordered copy (default) / batch as stored / no byte mirror
    if not labeled and batch.max_nodes <= 256:
        pins += ["blane", "blane-unordered"]               # adjacency bit-matrix mirror, lanes by expected length / as stored
    query = None
    if rng.integers(0, 3) == 0 and coo.G:                  # a query tail on a third of the rounds
        nc = np.maximum(np.asarray(d["node_counts"]), 1)
        query = np.stack([rng.integers(0, nc), rng.integers(0, nc)], 1).astype(np.int32)
    pad = bool(rng.integers(0, 4))
    K = int(rng.choice([1, 1, 2, 3, 7]))                    # epochs per launch (gtok_sent_params.epoch_count)
    if K * coo.G > 400000: K = 1
    u16 = bool(rng.integers(0, 2))                          # rows of 16-bit ids (GTOK_SENT_U16)
    ref = None
    for pin in pins:
        os.environ["GTOK_SENT_KERNEL"] = pin.split("-")[0]
        os.environ["GTOK_NO_LANE_SORT"] = "1" if pin == "lane-unsorted" else "0"
        os.environ["GTOK_NO_PACK8"] = "1" if pin == "lane-int32" else "0"
        os.environ["GTOK_BLANE_ORDER"] = "0" if pin == "blane-unordered" else "1"
        fresh = batch.to(DEV)                               # the resident layouts are made per pin
        ids, ln = gtok.ops.sent(fresh, nn, max_len, sd, ep, graph_base=base, query=None if query is None else torch.from_numpy(query), pad=pad,
                                epochs=K, u16=u16, **kw)
        ids, ln = ids.view(K, coo.G, -1), ln.view(K, coo.G)
        if ref is None:
            ref = [orc.sent(coo, nn, max_len, sd, ep + e, graph_base=base, ld=ids.shape[2], nthreads=T, query=query, **kw) for e in range(K)]
        for e in range(K):
            got = ids[e].cpu().numpy()
            got = got.view(np.uint16).astype(np.int32) if u16 else got
            inside = np.arange(ref[e][0].shape[1])[None, :] < ref[e][1][:, None] if not pad else np.ones_like(ref[e][0], bool)   # GTOK_SENT_NO_PAD: rows equal inside their lengths
            if not (np.array_equal(ln[e].cpu().numpy(), ref[e][1]) and np.array_equal(np.where(inside, got, 0), np.where(inside, ref[e][0], 0))):
                fails += 1; print(f"MISMATCH sent it={it} kind={kind} pin={pin or 'auto'} G={coo.G} max_len={max_len} query={query is not None} pad={pad} K={K} e={e} u16={u16}", flush=True)
        # the same launch with its rows packed as well (gtok_sent_packed where the lane kernel walks, else the one-pass pack behind the
        # walk), beside the slab or alone (GTOK_SENT_PACK_ONLY): re-padded through row_start, the rows are the oracle's
        if coo.G and rng.integers(0, 2):
            ldp = ids.shape[2]
            need = int(sum(int(((np.minimum(r[1], ldp) + 7) // 8 * 8).sum()) for r in ref))
            pk = gtok.ops.PackedRows(K * coo.G, int(need * 1.15) + 8 * 64 * 64, u16, DEV)
            alone = bool(rng.integers(0, 2))
            _, pln = gtok.ops.sent(fresh, nn, max_len, sd, ep, graph_base=base, query=None if query is None else torch.from_numpy(query), pad=pad,
                                   epochs=K, u16=u16, packed=pk, slab=not alone, ld=ldp, **kw)
            back = gtok.ops.unpack_rows_at(pk.buf, pk.row_start, pln.reshape(-1), ldp, 5, u16=u16).view(K, coo.G, ldp)
            for e in range(K):
                got = back[e].cpu().numpy()
                got = got.view(np.uint16).astype(np.int32) if u16 else got
                if int(pk.status()) or not (np.array_equal(pln.view(K, coo.G)[e].cpu().numpy(), ref[e][1]) and np.array_equal(got, ref[e][0])):
                    fails += 1; print(f"MISMATCH packed it={it} kind={kind} pin={pin or 'auto'} fused={pk.fused} alone={alone} G={coo.G} max_len={max_len} "
                                      f"query={query is not None} K={K} e={e} u16={u16} status={int(pk.status())}", flush=True)
            npacked = globals().get("npacked", 0) + 1; nfused = globals().get("nfused", 0) + int(bool(pk.fused))
    for k in ("GTOK_SENT_KERNEL", "GTOK_NO_LANE_SORT", "GTOK_NO_PACK8", "GTOK_BLANE_ORDER"):
        os.environ[k] = "" if k == "GTOK_SENT_KERNEL" else ("1" if k == "GTOK_BLANE_ORDER" else "0")
    if labeled:
        vocab = zinc_vocab(int(rng.integers(1, 45)), with_fallbacks=bool(rng.integers(0, 2)))
        lut = gtok.ops.zinc_lut(vocab, 45)
        ld = int(rng.choice([0, 8, 64, 241, 300]))
        iref = None
        for pin in ("", "quad", "lane", "wave"):
            os.environ["GTOK_IBTT_KERNEL"] = pin
            for gsz in ("8", "16"):
                os.environ["GTOK_IBTT_GROUP"] = gsz
                ids, ln = gtok.ops.ibtt_zinc(dev, lut, max_len, vocab["<pad>"], ld=ld or None)
                if iref is None:
                    iref = orc.ibtt_zinc(coo, lut.numpy(), max_len, vocab["<pad>"], ids.shape[1], nthreads=T)
                cmp(f"ibtt it={it} pin={pin or 'auto'} gs={gsz} G={coo.G} max_len={max_len} ld={ld}", ids, ln, *iref)
        os.environ["GTOK_IBTT_KERNEL"] = ""; os.environ["GTOK_IBTT_GROUP"] = ""
    else:
        vocab = {t: i for i, t in enumerate(["<pad>", "<bos>", "<e>", "<n>", "<q>", "<p>", "<eos>", "yes", "no", "has_cycle"] + [str(i) for i in range(nn)])}
        lut = gtok.ops.synth_lut(vocab, nn)
        q = np.zeros((coo.G, 4), np.int32); q[:, 0] = rng.integers(0, 4, coo.G); q[:, 1:] = rng.integers(0, len(vocab), (coo.G, 3))
        ids, ln = gtok.ops.ibtt_synth(dev, lut, torch.from_numpy(q), max_len, 0)
        cmp(f"ibtt_synth it={it} G={coo.G} max_len={max_len}", ids, ln, *orc.ibtt_synth(coo, lut.numpy(), q, max_len, 0, ids.shape[1], nthreads=T))
    if it % 5 == 0:
        print(f"it {it} done, {fails} mismatches, {time.time() - t0:.0f}s", flush=True)
print(f"packed launches {globals().get('npacked', 0)}, of them fused {globals().get('nfused', 0)}")
print("TOTAL mismatches", fails)
sys.exit(1 if fails else 0)
