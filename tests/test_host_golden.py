"""CPU: host-side mirror of the reference interface (strings, parsers, vocab builders, file loading, CSR
builder, C-ABI surface) against the golden vectors captured from the reference's own Python."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from _util import ROOT, config1_examples, edge_case_graphs, golden, golden2, golden_zinc_coo, gtok, zinc_data_list

gdl = gtok.graph_data_loader


def test_fixed_vocab_and_id_helpers():
    _, meta = golden()
    v, itos = gdl.build_fixed_zinc_vocab()
    assert list(v.items()) == [tuple(p) for p in meta["fixed_zinc_vocab"]]
    assert itos == {i: t for t, i in v.items()}
    assert gdl.SPECIAL == meta["special"]
    assert [gdl.get_atom_type_id(i) for i in range(9)] == meta["atom_ids"]
    assert [gdl.get_bond_type_id(i) for i in range(1, 5)] == meta["bond_ids"]
    assert list(gdl.get_zinc_num_types()) == meta["zinc_num_types"]
    for bad in (-1, 9, 100):
        with pytest.raises(ValueError):
            gdl.get_atom_type_id(bad)
    for bad in (0, 5):
        with pytest.raises(ValueError):
            gdl.get_bond_type_id(bad)
    assert gdl.get_atom_type_from_id(14) == "Cl" and gdl.get_bond_type_from_id(20) == "aromatic"
    for t, isn, ise, want in meta["map_autograph_token"]:
        try:
            got = gdl.map_autograph_token_to_fixed_id(t, 43, 52, is_node_type=bool(isn), is_edge_type=bool(ise))
        except ValueError:
            got = "ValueError"
        assert got == want, (t, isn, ise)
    ext = gdl.extend_vocab_with_dynamic_tokens(v, ["7", "C", "X", "7", "val_0_50"])
    assert (ext["7"], ext["X"], ext["val_0_50"], ext["C"]) == (22, 23, 24, 8) and len(ext) == 25


@pytest.mark.parametrize("max_len", [1024, 48])
def test_zinc_texts_match_reference(max_len):
    _, meta = golden()
    ds = gdl.ZINCTokenizationDataset(split="train", max_len=max_len, zinc_dataset=zinc_data_list(golden_zinc_coo()))
    items = [ds[i] for i in range(len(ds))]
    assert [it["text"] for it in items] == meta[f"zinc_L{max_len}_texts"]
    assert [it["label"] for it in items] == meta[f"zinc_L{max_len}_labels"]
    assert [it["graph_id"] for it in items] == meta[f"zinc_L{max_len}_graph_ids"]


def test_vocab_builders_match_reference():
    _, meta = golden()
    for c in meta["vocab_cases"]:
        v, itos = gdl.build_vocab_from_texts(c["texts"], min_freq=c["min_freq"], max_tokens=c["max_tokens"])
        assert list(v.items()) == [tuple(p) for p in c["vocab"]]
    for task in ("cycle_check", "shortest_path"):
        ex = meta[f"synth_{task}_examples"]
        v, _ = gdl.build_vocab_from_texts([e["text"] for e in ex], max_tokens=600)
        assert list(v.items()) == [tuple(p) for p in meta[f"synth_{task}_vocab"]]
        v40, _ = gdl.build_vocab_from_texts([e["text"] for e in ex], max_tokens=40)
        assert list(v40.items()) == [tuple(p) for p in meta[f"synth_{task}_vocab40"]]


def test_parsers_match_reference():
    _, meta = golden()
    out = iter(meta["parser_out"])
    for r in meta["parser_records"]:
        for task in ("cycle_check", "shortest_path"):
            want = next(out)
            edges, n, lab = gdl.parse_graph_from_json(r, task=task)
            q = gdl.graph_token_dataset_autograph.parse_query_nodes_from_text(r.get("text", ""))
            assert [list(map(int, e)) for e in edges] == want["edges"] and n == want["num_nodes"] and lab == want["label"]
            assert (None if q is None else list(q)) == want["query"]
    for task in ("cycle_check", "shortest_path"):
        for e, want in zip(meta[f"synth_{task}_examples"], meta[f"synth_{task}_parsed"]):
            edges, n, lab = gdl.parse_graph_from_json({"text": e["text"]}, task=task)
            assert [list(p) for p in edges] == want["edges"] and n == want["num_nodes"] and lab == want["label"]
            assert gdl.parse_label_from_text(e["text"], task) == want["label_from_text"]
            q = gdl.parse_query_nodes_from_text(e["text"])
            assert (None if q is None else list(q)) == want["query"]


def test_load_examples_matches_reference(tmp_path):
    """Same files on disk -> same example dicts as the reference's load_examples (order, labels, queries)."""
    _, meta = golden()
    for task in ("cycle_check", "shortest_path"):
        g = gtok.synth.graph_token_like(60, seed=1234, task=task)     # the corpus the fixture was made from
        for i, (txt, alg) in enumerate(zip(g["texts"], g["algorithms"])):
            dd = tmp_path / task / "tasks_train" / task / alg / "train"
            dd.mkdir(parents=True, exist_ok=True)
            (dd / f"{i:04d}.json").write_text(json.dumps([{"text": txt}]))
        ex = []
        for alg in sorted(set(g["algorithms"])):
            ex += gdl.load_examples(str(tmp_path / task / "tasks_train" / task / alg / "train" / "*.json"), task=task)
        assert ex == meta[f"synth_{task}_examples"]
        # the multi-algorithm wrapper walks the same directories
        allx = gdl.load_examples_multi_algorithm(str(tmp_path / task), task, sorted(set(g["algorithms"])), "train")
        assert allx == ex
        assert gdl.determine_num_classes(ex, task) == (2 if task == "cycle_check" else
                                                      max(e["label"] for e in ex if e["label"] is not None) + 1)
    # line-oriented / raw-text / malformed files, file sampling, per-file pair sampling, data_fraction
    d = tmp_path / "misc"; d.mkdir()
    for name, body in meta["loader_misc_files"].items():
        (d / name).write_text(body)
    for kw, want in zip(meta["loader_misc_calls"], meta["loader_misc_out"]):
        assert gdl.load_examples(str(d / "*.json"), **kw) == want, kw


_AGDS_SCRIPT = r"""
import importlib, json, sys
sys.path.insert(0, sys.argv[1])
gtok = importlib.import_module("glearning-benchmark_amd")
D = gtok.graph_data_loader.GraphTokenDatasetForAutoGraph
cases = json.load(open(sys.argv[3]))
out = []
for kw in cases:
    ds = D(sys.argv[2], **kw)
    out.append(dict(processed_dir=ds.processed_dir, raw_dir=ds.raw_dir, items=[
        dict(edge_index=d.edge_index.tolist(), edge_index_shape=list(d.edge_index.shape), y=d.y.tolist(), y_dtype=str(d.y.dtype),
             num_nodes=int(d.num_nodes), query_u=getattr(d, "query_u", None), query_v=getattr(d, "query_v", None)) for d in ds]))
json.dump(out, open(sys.argv[4], "w"))
"""


def test_graph_token_dataset_for_autograph_matches_reference(tmp_path):
    """GraphTokenDatasetForAutoGraph.process() (reference :259-408) item for item: file sampling per algorithm,
    num_pairs_per_graph sampling, val -> test fallback, INF / unlabeled / empty records skipped, query fields, the
    processed-cache key.  The per-algorithm sampling seed is `seed + hash(algo) % 10000` in the reference, so the
    mirror runs in a child interpreter with the PYTHONHASHSEED the fixture was generated under (0)."""
    import subprocess
    import sys
    _, meta = golden2()
    root = tmp_path / "graph-token"
    gtok.synth.write_tree(str(root), meta["agds_tree"])
    cases = tmp_path / "cases.json"; out = tmp_path / "out.json"
    cases.write_text(json.dumps([c["kwargs"] for c in meta["agds_cases"]]))
    env = dict(os.environ, PYTHONHASHSEED="0", PYTHONDONTWRITEBYTECODE="1")
    for attempt in ("processed from JSON", "served from the processed/ cache"):
        subprocess.run([sys.executable, "-c", _AGDS_SCRIPT, ROOT, str(root), str(cases), str(out)], check=True, env=env,
                       stdout=subprocess.DEVNULL)
        got = json.loads(out.read_text())
        assert len(got) == len(meta["agds_cases"])
        for g, want in zip(got, meta["agds_cases"]):
            assert os.path.relpath(g["processed_dir"], str(root)) == want["processed_dir"], attempt
            assert os.path.relpath(g["raw_dir"], str(root)) == want["raw_dir"]
            assert g["items"] == want["items"], (attempt, want["kwargs"])
            assert os.listdir(g["processed_dir"]) == ["data_gtok.pt"]      # our cache; never the reference's data.pt
    assert sum(len(c["items"]) for c in meta["agds_cases"]) > 100
    D = gdl.GraphTokenDatasetForAutoGraph
    with pytest.raises(RuntimeError) as e:
        D(str(root), task="cycle_check", algorithm=["nope"], split="train")
    assert str(e.value) == meta["agds_missing_error"]
    # the reference's own data.pt (a PyG pickle) in the same directory is neither read nor overwritten
    ds = D(str(root), task="cycle_check", algorithm=["er", "ba", "path"], split="train")
    path = ds.processed_paths[0]
    assert path.endswith("data.pt") and not os.path.exists(path)
    with open(path, "wb") as f:
        f.write(b"not a gtok cache")
    ds2 = D(str(root), task="cycle_check", algorithm=["er", "ba", "path"], split="train")
    assert len(ds2) == len(ds) and open(path, "rb").read() == b"not a gtok cache"
    b = ds2.graph_batch()
    assert b.num_graphs == len(ds2) and int(b.node_ptr[-1]) == sum(d.num_nodes for d in ds2)


@pytest.mark.parametrize("task", ["cycle_check", "shortest_path"])
def test_config1_corpus_host_side(task, tmp_path):
    """BASELINE config 1 at 1k records: our generator's tree -> the mirror's loader gives the example dicts the
    reference's loader gave (texts, labels, queries), and the frequency-ordered vocab is the reference's."""
    arr, meta = golden2()
    algs = ["er", "ba", "sbm", "path", "star", "complete"]
    gtok.synth.write_tree(str(tmp_path), gtok.synth.graph_token_tree(167, seed=1234, task=task, algorithms=algs, splits=("train",)))
    ex = gdl.load_examples_multi_algorithm(str(tmp_path), task, algs, "train", seed=0)
    want = config1_examples(task)
    assert len(ex) == 1002 and ex == want
    v, _ = gdl.build_vocab_from_texts([e["text"] for e in ex], max_tokens=600)
    assert list(v.items()) == [tuple(p) for p in meta[f"config1_{task}_vocab"]]


def test_host_collates_match_reference():
    arr, meta = golden()
    ids, ln = arr["zinc_L1024_ids"], arr["zinc_L1024_len"]
    batch = [(torch.tensor(ids[i, :ln[i]]), torch.tensor(int(arr["zinc_L1024_y"][i]))) for i in range(16)]
    X, A, Y = gdl.collate(batch, 2)
    assert X.dtype == torch.int64 and A.dtype == torch.bool and Y.dtype == torch.int64
    assert np.array_equal(X.numpy(), arr["zinc_L1024_collate_X"]) and np.array_equal(A.numpy(), arr["zinc_L1024_collate_A"])
    assert np.array_equal(Y.numpy(), arr["zinc_L1024_collate_Y"])
    out, oln = arr["agtt_remap_out"], arr["agtt_remap_len"]
    items = [(torch.tensor(out[i, :oln[i]]), torch.ones(int(oln[i]), dtype=torch.bool), meta["agtt_zinc_labels"][i], None)
             for i in range(16)]
    X, A, Y, dl = gtok.agtt.collate_fn(items)
    assert np.array_equal(X.numpy(), arr["agtt_zinc_collate_X"]) and np.array_equal(A.numpy(), arr["agtt_zinc_collate_A"])
    assert str(Y.dtype) == meta["agtt_zinc_collate_Y_dtype"] and np.allclose(Y.numpy(), arr["agtt_zinc_collate_Y"])
    items = [(torch.tensor(arr["agtt_sp_out"][i, :arr["agtt_sp_out_len"][i]]),
              torch.ones(int(arr["agtt_sp_out_len"][i]), dtype=torch.bool), int(arr["agtt_sp_labels"][i]), None) for i in range(16)]
    X, A, Y, dl = gtok.agtt.collate_fn(items)
    assert np.array_equal(X.numpy(), arr["agtt_sp_collate_X"]) and str(Y.dtype) == meta["agtt_sp_collate_Y_dtype"]
    assert np.array_equal(Y.numpy(), arr["agtt_sp_collate_Y"])


def test_tokenizer_object_interface():
    T = gtok.Graph2TrailTokenizer
    assert (T.sos, T.reset, T.ladj, T.radj, T.eos, T.pad) == (0, 1, 2, 3, 4, 5)
    t = T(dataset_names=[], max_length=1024, truncation_length=1024, labeled_graph=True, undirected=True)
    assert t.idx_offset == 6
    with pytest.raises(RuntimeError):
        t.set_num_node_and_edge_types(9, 4)
    t.set_num_nodes(37); t.set_num_node_and_edge_types(9, 4)
    assert (t.node_idx_offset, t.edge_idx_offset) == (43, 52)
    u = T(dataset_names=[], max_length=600, truncation_length=600, labeled_graph=False, undirected=True)
    u.set_num_nodes(49)
    assert u.idx_offset + 49 + 1 == 56          # vocab_size formula of train_agtt.py:586
    with pytest.raises(ValueError):
        T(dataset_names=["zinc"])


def test_csr_builder():
    d = edge_case_graphs()
    b = gtok.GraphBatch.from_coo(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"])
    nptr, eptr = b.node_ptr.numpy(), b.edge_ptr.numpy()
    assert b.num_graphs == len(d["node_counts"]) and b.max_nodes == 8 and b.max_edges == 56
    assert b.eorder is not None                      # Mol B & co. are not row-sorted
    for g in range(b.num_graphs):
        n, e0, e1 = nptr[g + 1] - nptr[g], eptr[g], eptr[g + 1]
        rp = b.rowptr.numpy()[nptr[g] + g: nptr[g] + g + n + 1]
        assert rp[0] == 0 and rp[-1] == e1 - e0 and (np.diff(rp) >= 0).all()
        src, dst = d["src"][e0:e1], d["dst"][e0:e1]
        for u in range(n):
            ks = np.arange(rp[u], rp[u + 1])
            orig = b.eorder.numpy()[e0 + ks]
            assert (src[orig] == u).all() and (dst[orig] == b.col.numpy()[e0 + ks]).all()
            assert (np.diff(orig) > 0).all()          # stable: original order kept inside a row
            assert (b.eattr.numpy()[e0 + ks] == np.minimum(d["edge_attr"][e0:e1][orig], 255)).all()
    z = gtok.synth.zinc_like(50, seed=1)
    bz = gtok.GraphBatch.from_coo(z["node_counts"], z["edge_counts"], z["src"], z["dst"], z["x"], z["edge_attr"])
    assert bz.eorder is None                          # coalesced input: identity order is not stored
    sh = bz.shard(10, 30)
    assert sh.num_graphs == 20 and int(sh.edge_ptr[-1]) == sh.col.numel() and int(sh.node_ptr[0]) == 0
    with pytest.raises(ValueError):
        gtok.GraphBatch.from_coo([2], [1], [0], [2])
    dl = gtok.GraphBatch.from_data_list(zinc_data_list(golden_zinc_coo()))
    g = golden_zinc_coo()
    ref = gtok.GraphBatch.from_coo(g["node_counts"], g["edge_counts"], g["src"], g["dst"], g["x"], g["edge_attr"])
    assert torch.equal(dl.col, ref.col) and torch.equal(dl.rowptr, ref.rowptr) and torch.equal(dl.eattr, ref.eattr)


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "gtok.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char \*)\s*(gtok_\w+)\(", hdr, flags=re.M))
    assert declared == set(gtok._lib.SYMBOLS), declared ^ set(gtok._lib.SYMBOLS)
    lib = ctypes.CDLL(gtok._lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert gtok.lib().gtok_version() == gtok._lib.ABI_VERSION and gtok.lib().gtok_target() == b"gfx950"


def test_product_has_no_cpu_path():
    z = gtok.synth.zinc_like(4, seed=0)
    b = gtok.GraphBatch.from_coo(z["node_counts"], z["edge_counts"], z["src"], z["dst"], z["x"], z["edge_attr"])
    with pytest.raises(gtok.GtokError):
        gtok.ops.sent(b, 37, 1024, 0)                # host tensors: refused, never silently tokenized on the CPU
    with pytest.raises(gtok.GtokError):
        gtok.ops.ibtt_zinc(b, torch.zeros(64, dtype=torch.int32), 1024, 2)
    if not torch.cuda.is_available():
        with pytest.raises(gtok.GtokError):
            gdl.TokenDataset([{"text": "<bos> yes", "label": 1}], {t: i for i, t in enumerate(gdl.SPECIAL)})
    # nothing under the product package imports, links or loads the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "glearning-benchmark_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                for needle in ("import oracle", "from oracle", "libgtok_oracle", "oracle.py", "#include \"../../oracle"):
                    assert needle not in src, (f, needle)


def test_custom_ops_are_registered_cuda_only_and_have_meta_kernels():
    for name in ("sent", "ibtt_zinc", "ibtt_synth", "remap_zinc", "collate", "text_to_ids", "find_token",
                 "vocab_stats_synth", "parse_graph_text", "sent_decode", "row_offsets", "pack_rows", "unpack_rows", "collate_packed"):
        assert hasattr(torch.ops.gtok, name), name
    x = torch.zeros((3, 7), dtype=torch.int64)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.gtok.find_token(x, 4)
    assert torch.ops.gtok.find_token(x.to("meta"), 4).shape == (3,)
    dec = torch.ops.gtok.sent_decode(torch.zeros((5, 16), dtype=torch.int32, device="meta"),
                                     torch.zeros(5, dtype=torch.int32, device="meta"), 37, True, 9, 12, 37)
    assert [tuple(t.shape) for t in dec] == [(5,), (5,), (5,), (5, 12), (5, 12), (5, 12), (5, 37)]
    ids = torch.zeros((4, 8), dtype=torch.int32); ln = torch.full((4,), 8, dtype=torch.int32)
    with pytest.raises((NotImplementedError, RuntimeError)):       # no CPU kernel exists
        torch.ops.gtok.remap_zinc(ids, ln, 6, 43, 52)
    out = torch.ops.gtok.remap_zinc(ids.to("meta"), ln.to("meta"), 6, 43, 52)   # shape inference without a GPU
    assert out.shape == (4, 8) and out.device.type == "meta"
    ptr = torch.ops.gtok.row_offsets(ln.to("meta"), 8, 8)
    packed, st = torch.ops.gtok.pack_rows(ids.to("meta"), ln.to("meta"), ptr, 2, 64)
    assert ptr.shape == (5,) and packed.shape == (64,) and packed.dtype == torch.int16 and st.shape == (1,)
    assert torch.ops.gtok.unpack_rows(packed, ptr, ln.to("meta"), 8, 5, 0, 0).shape == (4, 8)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.gtok.row_offsets(ln, 8, 8)


def test_bench_spawns_its_own_ranks_before_touching_the_gpu():
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset) starts its N ranks itself - children, started before
    the parent initialises HIP - relays rank 0's line and fails loudly when a rank fails.  Without a GPU every rank exits
    with bench.py's own "needs a GPU" message: the parent must report exactly that, non-zero."""
    import subprocess
    import sys
    if __import__("torch").cuda.is_available():
        pytest.skip("covers the GPU-less failure path")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "failed with exit code" in r.stderr and "needs a GPU" in r.stderr, r.stderr[-1500:]


def test_bench_ends_every_rank_when_one_fails_while_rank_0_would_wait(tmp_path):
    """A rank other than 0 that dies early (bad device, import error) must not leave rank 0 in the rendezvous until the
    process-group timeout: spawn_ranks polls all children, terminates the rest on the first non-zero exit and reports the
    failed rank's stderr (ADVICE r4 / VERDICT r4 #5)."""
    import subprocess
    import sys
    import time as _time
    script = tmp_path / "rank.py"
    marker = tmp_path / "rank0.pid"
    script.write_text(
        "import os, sys, time\n"
        "r = int(os.environ['RANK'])\n"
        "if r == 1:\n"
        "    sys.stderr.write('rank 1: no such device\\n'); sys.exit(3)\n"
        f"open({str(marker)!r}, 'w').write(str(os.getpid()))\n"
        "time.sleep(600)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = _time.time()
    r = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); import bench; bench.spawn_ranks(2, argv=[], script={str(script)!r})"],
                       capture_output=True, text=True, timeout=240, env=env)
    took = _time.time() - t0
    assert r.returncode != 0 and "rank 1 failed with exit code 3" in r.stderr and "no such device" in r.stderr, r.stderr[-1500:]
    assert took < 120, f"the parent waited {took:.0f} s for a rank that was never going to finish"
    if marker.exists():                                 # rank 0 got as far as writing its pid: it must be gone now
        pid = int(marker.read_text())
        gone = False
        for _ in range(50):
            try:
                os.kill(pid, 0)
            except OSError:
                gone = True
                break
            _time.sleep(0.1)
        assert gone, "rank 0 is still running after the parent returned"


def test_raw_tensor_op_cache_finds_live_batches_and_drops_dead_ones():
    """torch.ops.gtok.sent without prepared arrays keeps the batch it prepares per set of tensor objects (identity + version
    counters).  The cache must find a live batch again, miss after an in-place write, and not keep a corpus alive that only the
    cache itself still references (host logic: no GPU needed)."""
    import gc
    to = gtok.torch_ops
    to._BATCHES.clear()
    mk = lambda n: (torch.zeros(n + 1, dtype=torch.int32), torch.zeros(n + 1, dtype=torch.int64), torch.zeros(n, dtype=torch.int32), torch.zeros(n, dtype=torch.int32))
    a, c = mk(5), mk(6)
    b1 = to._batch(*a, None, None, None, 3, 3)
    assert to._batch(*a, None, None, None, 3, 3) is b1 and len(to._BATCHES) == 1
    assert to._batch(*a, None, None, None, 4, 3) is not b1            # other maxima: another batch
    b2 = to._batch(*c, None, None, None, 3, 3)
    n_before = len(to._BATCHES)
    del b1, a
    gc.collect()
    d = mk(7)
    to._batch(*d, None, None, None, 3, 3)                              # a miss purges what nobody else holds
    assert len(to._BATCHES) == n_before - 2 + 1                        # both entries of `a` gone, `d` added
    c[3].add_(1)                                                       # written in place: not the batch that was prepared
    assert to._batch(*c, None, None, None, 3, 3) is not b2
    to._BATCHES.clear()
