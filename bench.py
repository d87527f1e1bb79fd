#!/usr/bin/env python
"""bench.py — graphs tokenized/sec + emitted tokens/sec on the ZINC-full-shaped corpus (BASELINE.json).

A "step" is one epoch pass of the AGTT hot path over the rank's resident corpus: labelled SENT trail
walk fused with the fixed-vocab ZINC remap (trainer/train_agtt.py:246-254), CSR already in HBM.
One process per GPU; ranks hold their own 249,456-graph shard (weak scaling, no data-path
collective; the optional all-gather of the padded slab is timed separately and reported beside it).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
gtok = importlib.import_module("glearning-benchmark_amd")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
ZINC_FULL_GRAPHS = 249456      # 220,011 / 24,445 / 5,000
WORKLOADS = {
    "zinc_full": dict(graphs=ZINC_FULL_GRAPHS, desc="AGTT labelled SENT + fused ZINC remap, ZINC-full-shaped synthetic molecules"),
    "zinc_subset": dict(graphs=12000, desc="AGTT labelled SENT + fused ZINC remap, ZINC-subset-shaped synthetic molecules"),
    # BASELINE config 5 shape (graph_generator.sh families, 10..256 nodes, sparsity 0.1-0.2, max_len 600):
    # 125k graphs = one rank's share of the 1M-graph corpus on 8 GPUs (sampled on the GPU in ~6 s)
    "synth_er": dict(graphs=125000, desc="AGTT unlabelled SENT, Erdos-Renyi graph-token-shaped graphs, 10-256 nodes, max_len 600"),
    # the same share of config 5 with the full family mix of graph_generator.sh (er/ba/sbm/sfn/path/star/complete, graph i of
    # family i mod 7): a seventh of the graphs are complete - they hold most of the corpus's entries
    "synth_mix": dict(graphs=125000, desc="AGTT unlabelled SENT, er/ba/sbm/sfn/path/star/complete graph-token-shaped graphs, 10-256 nodes, max_len 600"),
}


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(msg, file=sys.stderr, flush=True)


def event_ms(pairs):
    return [s.elapsed_time(e) for s, e in pairs]


def host_cores(limit):
    """Threads the cpu_baseline leg may really run on: the affinity mask, cut to the cgroup's CPU quota (a GPU box hands
    out 16 CPUs' worth of time on a 256-thread host: 128 OpenMP threads under that quota are throttled, and the figure
    moved by 2.5 x between runs), cut to what the oracle's OpenMP runtime offers."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(q) // int(per)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = int(f.read()), int(g.read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, min(n, limit))


def timed_loop(fn, steps, multi, per_launch_events=True):
    """barrier + synchronize on both sides, K launches between, HIP events recorded on the stream the kernels run on.
    per_launch_events=True: an event pair around EVERY launch (the kernel's own duration; each pair costs a few
    microseconds of idle stream between launches).  False: ONE pair around the K launches - the timed region of the
    headline: returns K copies of (elapsed / K), i.e. the average launch-to-launch time, gaps included."""
    n_ev = steps if per_launch_events else 1
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if not per_launch_events:
        pairs[0][0].record()
    for k in range(steps):
        if per_launch_events:
            pairs[k][0].record()
        fn(k)
        if per_launch_events:
            pairs[k][1].record()
    if not per_launch_events:
        pairs[0][1].record()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = event_ms(pairs)
    return wall, (ms if per_launch_events else [ms[0] / steps] * steps)


def boundary_section(d, G, dev, max_nodes, max_len):
    """Throughput THROUGH the drop-in boundary, i.e. what the reference's call sites see (items/s; outside the timed
    region; rank 0, N=1).  The corpus is handed over the way torch_geometric hands ZINC over: an InMemoryDataset-like
    object with collated storage (synth.InMemoryLike).  Python per-item work (fetching an item object, slicing a row)
    bounds every per-item call site at ~10^5 items/s whatever the kernels do; device_batches / tokenize are the
    batched routes."""
    import contextlib
    gdl = gtok.graph_data_loader
    out = {}
    with contextlib.redirect_stdout(sys.stderr):             # the dataset classes print like the reference's do
        pyg = gtok.synth.InMemoryLike(d)
        sync = torch.cuda.synchronize

        import gc

        def clock(f):
            # (a full collection BEFORE the clock starts: a leg then runs against the young generations only - the corpus objects of the
            # earlier legs, millions of them, made a generation-2 pass inside a 9 ms leg cost several times the leg)
            gc.collect()
            sync(); t0 = time.perf_counter(); r = f(); sync()
            return time.perf_counter() - t0, r
        # CSR ingestion (once per split)
        gtok.GraphBatch.from_dataset(gtok.synth.InMemoryLike(gtok.synth.zinc_like(2000, seed=1)), device=dev)      # torch warm-up
        t, _ = clock(lambda: gtok.GraphBatch.from_dataset(pyg, device=dev))
        S = min(G, 50000)
        t_fetch, items = clock(lambda: [pyg[i] for i in range(S)])
        t_list, _ = clock(lambda: gtok.GraphBatch.from_data_list(items))
        out["csr_ingestion"] = dict(collated_storage_graphs_per_sec=round(G / t, 1), collated_storage_seconds=round(t, 4), graphs=G,
                                    item_list_graphs_per_sec=round(S / t_list, 1), item_list_sample=S,
                                    item_fetch_graphs_per_sec=round(S / t_fetch, 1),
                                    note="collated: GraphBatch.from_dataset on InMemoryDataset-style storage, built on the device; "
                                         "item list: GraphBatch.from_data_list over already fetched items (host); item fetch = the stand-in's __getitem__")
        del items

        def tokenizer():
            tok = gtok.Graph2TrailTokenizer(dataset_names=[], max_length=max_len, truncation_length=max_len, labeled_graph=True,
                                            undirected=True, device=dev)
            tok.set_num_nodes(max_nodes); tok.set_num_node_and_edge_types(*gdl.get_zinc_num_types())
            return tok
        # AGTT, zero-edit: the reference's own __getitem__ calls tokenizer(data) per item (train_agtt.py:246-250)
        src = gdl.ZINCDatasetForAutoGraph(split="train", zinc_dataset=pyg)
        tok = tokenizer()

        def per_item_loop():
            n = 0
            for i in range(G):
                n += tok(src[i]).numel()
            return n
        t1, _ = clock(per_item_loop)                           # epoch 0: includes ingestion + the first launch
        t2, ntok = clock(per_item_loop)                        # epoch 1: steady state
        out["agtt_zero_edit_tokenizer_call"] = dict(items_per_sec=round(G / t2, 1), first_epoch_items_per_sec=round(G / t1, 1),
                                                    launches_per_epoch=tok.launches / 2, epochs_per_launch=tok.epochs_for(G),
                                                    tokens_per_sec=round(ntok / t2, 1), items=G,
                                                    note="for i: tokens = tokenizer(pyg_dataset[i]) with this package's ZINCDatasetForAutoGraph: "
                                                         "one gtok_sent launch per K epochs (16-bit rows) + one packed int64 D2H copy per epoch, rows are slices of it")
        S = min(G, 2000)
        loose = [pyg[i] for i in range(S)]                     # objects the datasets did not hand out: one launch per call
        t, _ = clock(lambda: [tok(x) for x in loose])
        out["agtt_untagged_tokenizer_call"] = dict(items_per_sec=round(S / t, 1), sample=S,
                                                   note="tokenizer(data) on foreign objects: CSR of one graph + H2D + launch + D2H per item")
        # AGTT, one-line swap: agtt.TokenizedGraphDataset.__getitem__ over a full epoch, and device_batches
        ds = gtok.agtt.TokenizedGraphDataset(src, tokenizer(), task="zinc", remap_to_fixed_vocab=True, device=dev)

        def getitem_loop():
            n = 0
            for i in range(G):
                n += ds[i][0].numel()
            return n
        clock(getitem_loop)
        t, _ = clock(getitem_loop)
        out["agtt_dataset_getitem"] = dict(items_per_sec=round(G / t, 1), items=G,
                                           note="agtt.TokenizedGraphDataset.__getitem__ over a whole epoch (fused remap; incl. the packed D2H copy)")

        # the reference's loader construction (train_agtt.py:599-607) unchanged over the one-line-swap class: the stock DataLoader
        # fetches whole batches through __getitems__ (collated on the device, collate_fn passes them through)
        from torch.utils.data import DataLoader

        def loader_epoch(dl):
            n = 0
            for X, A, Y, data_list in dl:
                n += X.shape[0]
            return n
        for shuffle in (True, False):
            dl = DataLoader(ds, batch_size=128, shuffle=shuffle, num_workers=0, collate_fn=gtok.agtt.collate_fn)
            clock(lambda: loader_epoch(dl))
            t, n = clock(lambda: loader_epoch(dl))
            out["agtt_dataloader_shuffle" if shuffle else "agtt_dataloader"] = dict(
                items_per_sec=round(n / t, 1), batch_size=128, items=n, shuffle=shuffle,
                note="for X, A, Y, data_list in DataLoader(agtt.TokenizedGraphDataset, batch_size=128, shuffle, num_workers=0, collate_fn=agtt.collate_fn): "
                     "batch-level fetch (__getitems__ -> gtok_collate_packed on the 16-bit slab), data_list is a lazy sequence")

        # the same loader with this package's batch sampler in place of `shuffle=True` (one permutation per epoch, cut into lists): the
        # stock RandomSampler -> BatchSampler chain costs ~38 us per batch of 128 on this host before the dataset is asked for anything
        dl = DataLoader(ds, batch_sampler=gtok.agtt.EpochBatchSampler(len(ds), 128, shuffle=True), num_workers=0, collate_fn=gtok.agtt.collate_fn)
        clock(lambda: loader_epoch(dl))
        t, n = clock(lambda: loader_epoch(dl))
        out["agtt_dataloader_batch_sampler"] = dict(items_per_sec=round(n / t, 1), batch_size=128, items=n, shuffle=True,
                                                    note="DataLoader(ds, batch_sampler=agtt.EpochBatchSampler(len(ds), 128, shuffle=True), num_workers=0, collate_fn=agtt.collate_fn)")
        # ... and with the sampler announcing its batches to the dataset: one gtok_collate_epoch call per epoch, a batch is three views
        dl = DataLoader(ds, batch_sampler=gtok.agtt.EpochBatchSampler(len(ds), 128, shuffle=True, dataset=ds), num_workers=0, collate_fn=gtok.agtt.collate_fn)
        clock(lambda: loader_epoch(dl))
        t, n = clock(lambda: loader_epoch(dl))
        out["agtt_dataloader_planned"] = dict(items_per_sec=round(n / t, 1), batch_size=128, items=n, shuffle=True,
                                              note="DataLoader(ds, batch_sampler=agtt.EpochBatchSampler(len(ds), 128, shuffle=True, dataset=ds), num_workers=0, "
                                                   "collate_fn=agtt.collate_fn): the loader's two lines of trainer/train_agtt.py:599-601 with the sampler swapped")

        def batches(with_data, bs=128):
            n = 0
            for X, A, Y, dl in ds.device_batches(bs, epoch=7, with_data=with_data):
                n += X.shape[0]
            return n
        clock(lambda: batches(True))
        # (median of three epochs: an epoch is ~9 ms of host work, and one allocator or collector hiccup in it halved single samples)
        t, n = sorted(clock(lambda: batches(True)) for _ in range(3))[1]
        t_nd, _ = sorted(clock(lambda: batches(False)) for _ in range(3))[1]
        out["agtt_device_batches"] = dict(items_per_sec=round(n / t, 1), items_per_sec_without_data_list=round(n / t_nd, 1), batch_size=128, items=n, epochs_timed=3,
                                          note="agtt.TokenizedGraphDataset.device_batches(128): the whole epoch collated by one gtok_collate_epoch call, batches are views; X / attn / labels stay on the device; "
                                               "the list of Data objects collate_fn returns is fetched item by item unless with_data=False")
        # IBTT: strings -> TokenDataset (train_ibtt.py:229-235, :395-397) and the string-free route
        zds = gdl.ZINCTokenizationDataset(split="train", max_len=max_len, zinc_dataset=pyg)
        t_str, ex = clock(lambda: [zds[i] for i in range(G)])          # first fetch renders the whole split on the device
        vocab, _ = gdl.build_fixed_zinc_vocab()
        t_voc, vocab = clock(lambda: gdl.build_zinc_vocab_on_device([e["text"] for e in ex], device=dev))
        t_td, td = clock(lambda: gdl.TokenDataset(ex, vocab, max_len, device=dev))
        out["ibtt_strings"] = dict(items_per_sec=round(G / t_str, 1), seconds=round(t_str, 3), items=G,
                                   note="[ds[i] for i in range(n)] (train_ibtt.py:229-235): ZINCTokenizationDataset renders the split's strings "
                                        "with gtok_ibtt_zinc + gtok_ids_to_text on the first fetch, items are served from that")
        out["ibtt_vocab_scan"] = dict(items_per_sec=round(G / t_voc, 1), seconds=round(t_voc, 3), vocab_size=len(vocab),
                                      note="build_zinc_vocab_on_device(texts): the dynamic-token scan of train_ibtt.py:361-372 (gtok_vocab_stats_text)")
        out["ibtt_token_dataset_init"] = dict(items_per_sec=round(G / t_td, 1), seconds=round(t_td, 3), items=G,
                                              note="TokenDataset(examples, vocab, max_len): pack texts + gtok_text_to_ids + one packed host copy of the rows")
        out["ibtt_examples_to_token_dataset_seconds"] = round(t_str + t_td, 3)
        pad_id = vocab["<pad>"]
        dl = DataLoader(td, batch_size=128, shuffle=True, num_workers=0, collate_fn=lambda b: gdl.collate(b, pad_id))
        clock(lambda: sum(X.shape[0] for X, A, Y in dl))
        t, n = clock(lambda: sum(X.shape[0] for X, A, Y in dl))
        out["ibtt_dataloader"] = dict(items_per_sec=round(n / t, 1), batch_size=128, items=n,
                                      note="for X, attn, Y in DataLoader(TokenDataset, batch_size=128, shuffle=True, collate_fn=lambda b: collate(b, pad_id)) "
                                           "(train_ibtt.py:399-402; num_workers=0 here): batches gathered from the packed host buffer (__getitems__)")
        S = min(G, 5000)
        zds._bulk = False; zds._texts = None
        t_py, _ = clock(lambda: [zds[i] for i in range(S)])
        zds._bulk = True
        out["ibtt_strings_per_item_python"] = dict(items_per_sec=round(S / t_py, 1), sample=S, note="the restated per-item Python renderer (no GPU: host tests)")
        t1, _ = clock(lambda: zds.tokenize(vocab, max_len, device=dev))
        t2, _ = clock(lambda: zds.tokenize(vocab, max_len, device=dev))
        out["ibtt_tokenize_csr"] = dict(first_call_graphs_per_sec=round(G / t1, 1), resident_graphs_per_sec=round(G / t2, 1), graphs=G,
                                        note="ZINCTokenizationDataset.tokenize(vocab): CSR -> ids, no strings (first call includes ingestion)")
    return out


def spawn_ranks(n, argv=None, script=None, poll_s=0.2):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves - fresh child processes, started BEFORE
    this process has touched the GPU (nothing above this line initialises HIP), one per GPU, rendezvous on 127.0.0.1 -
    relay rank 0's JSON line and exit non-zero if any rank failed.  All children are polled: as soon as ONE exits non-zero
    the others are terminated (a rank that died before or inside the rendezvous would otherwise leave its siblings in
    init_process_group / a collective until the process-group timeout, holding their GPUs) and the run ends with that rank's
    exit code and the tail of its stderr."""
    import socket
    import subprocess
    import tempfile
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    argv = sys.argv[1:] if argv is None else list(argv)
    script = os.path.abspath(__file__) if script is None else script
    procs, errs = [], []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        errs.append(tempfile.TemporaryFile())
        procs.append(subprocess.Popen([sys.executable, script] + argv, env=env, stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=errs[-1]))
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(poll_s)
    if failed is not None:
        for p in procs:                      # the exact children started above, nothing matched by name
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill(); p.wait()
    for r, f in enumerate(errs):             # children's stderr: relayed in rank order once they are done
        f.seek(0)
        data = f.read().decode(errors="replace")
        if failed is not None and r == failed[0]:
            sys.stderr.write(f"[bench] rank {r} exited with code {failed[1]}; the last lines of its stderr:\n" + "\n".join(data.splitlines()[-25:]) + "\n")
        elif failed is None:
            sys.stderr.write(data)
    sys.stderr.flush()
    if failed is not None:
        raise SystemExit(f"bench.py: rank {failed[0]} failed with exit code {failed[1]}; the other ranks were terminated")
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()


def main():
    # ONE JSON line on stdout: native libraries write there too (RCCL prints its version banner at init), so file
    # descriptor 1 points at stderr for the whole run and the line goes out through a private copy of the real stdout
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="zinc_full", choices=sorted(WORKLOADS))
    ap.add_argument("--graphs", type=int, default=None, help="graphs per rank (default: the workload's size)")
    ap.add_argument("--ld", default="tight", choices=["tight", "safe"],
                    help="slab width: measured max length + margin (verified after the run) or the a-priori bound")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank tokenizes its own ZINC-full-sized corpus (headline; no data-path collective). "
                         "strong: one corpus block-sharded over the ranks + all-gather (BASELINE config 4) becomes the headline; "
                         "multi-rank runs report it beside the weak line either way")
    ap.add_argument("--cpu-sample", type=int, default=None, help="graphs in the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ibtt", action="store_true")
    ap.add_argument("--rows", default="padded", choices=["padded", "unpadded", "u16", "u16padded"],
                    help="the flavour the timed step itself runs in (counter passes of that flavour; the headline stays 'padded'): "
                         "unpadded = GTOK_SENT_NO_PAD, u16 = GTOK_SENT_U16 | GTOK_SENT_NO_PAD (16-bit rows, tokens only), u16padded = GTOK_SENT_U16")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 1 s back-to-back leg")
    ap.add_argument("--no-boundary", action="store_true", help="skip the boundary section (call-site throughput through the Dataset classes)")
    ap.add_argument("--no-unpadded", action="store_true", help="skip the GTOK_SENT_NO_PAD leg (profiling runs: one launch flavour per kernel name)")
    ap.add_argument("--epochs-per-launch", type=int, default=None,
                    help="epochs (steps) one gtok_sent launch carries (gtok_sent_params.epoch_count). Default: 1 for corpora that fill "
                         "the chip on their own (zinc_full, synth_*), 24 for zinc_subset - the trainer re-tokenizes its split "
                         "every epoch and trails depend on (seed, epoch, graph) only")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        os.dup2(real_stdout, 1)
        return spawn_ranks(args.gpus)

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the tokenizer has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # GTOK_BENCH_FORCE_DIST=1 runs the collective legs even with one rank (rehearsal of the N>1 path on a 1-GPU box)
    multi = world > 1 or os.environ.get("GTOK_BENCH_FORCE_DIST") == "1"
    if multi:
        import datetime
        # a short rendezvous / collective timeout: a rank that never arrives must fail this run in minutes, not in the default 10
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=int(os.environ.get("GTOK_BENCH_PG_TIMEOUT", "180"))))

    wl = WORKLOADS[args.workload]
    G = args.graphs or wl["graphs"]
    zinc = args.workload.startswith("zinc")
    max_len, ntypes, etypes = (1024, 9, 4) if zinc else (600, 0, 0)
    t_gen = time.perf_counter()
    if zinc:
        d = gtok.synth.zinc_like(G, seed=1000 + rank)
        host = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)
    else:
        sample = gtok.synth.mix_batch_device if args.workload == "synth_mix" else gtok.synth.er_batch_device
        d = sample(G, dev, seed=1000 + rank)
        host = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], device=dev)
    batch = host          # CSR built on the device (torch sort / bincount): bit-identical to the host builder, much faster
    torch.cuda.synchronize()
    log(f"[bench] rank corpus: {G} graphs, {host.num_nodes_total} nodes, {host.num_edges_total} CSR entries "
        f"({time.perf_counter() - t_gen:.1f}s to generate + upload)")
    max_nodes = gtok.dist.all_reduce_max_int(host.max_nodes, dev)   # tokenizer.set_num_nodes (train_agtt.py:534)
    graph_base = rank * G
    kw = dict(labeled=zinc, num_node_types=ntypes, num_edge_types=etypes, remap_zinc=zinc, graph_base=graph_base)

    # slab width
    safe_ld = gtok.ops.sent_safe_ld(batch, zinc, max_len)
    if args.ld == "tight":
        _, ln0 = gtok.ops.sent(batch, max_nodes, max_len, seed=0, epoch=0, ld=safe_ld, **kw)
        ld = min(safe_ld, (int(ln0.max().item()) * 5 // 4 + 8 + 15) // 16 * 16)   # rows start on 64-byte boundaries
        ld = gtok.dist.all_reduce_max_int(ld, dev)
    else:
        ld = gtok.dist.all_reduce_max_int(safe_ld, dev)
    # epochs per launch: a step is one epoch pass; one gtok_sent launch may carry E of them (gtok_sent_params.epoch_count: the
    # trainer re-tokenizes the same split every epoch, trainer/train_agtt.py:246-250, :676-680, and a trail is a pure function
    # of (seed, epoch, graph)) - a split that cannot fill the chip on its own (zinc_subset: 12 k molecules) is launched 24 epochs
    # at a time.  K steps = ceil(K / E) launches, the last one with the remaining epochs.
    E = args.epochs_per_launch or (24 if args.workload == "zinc_subset" else 1)
    E = max(1, min(E, args.steps))
    n_launch = -(-args.steps // E)
    rows_u16, rows_pad = args.rows.startswith("u16"), args.rows in ("padded", "u16padded")
    ids = torch.empty((E * G, ld), dtype=torch.int16 if rows_u16 else torch.int32, device=dev)
    lens_all = torch.empty((n_launch * E, G), dtype=torch.int32, device=dev)
    lens = [lens_all[k] for k in range(args.steps)]
    scratch_len = torch.empty((E * G,), dtype=torch.int32, device=dev)

    def step(k, ln):          # ONE epoch in one launch (secondary legs and the end-of-run parity check)
        gtok.ops.sent(batch, max_nodes, max_len, seed=0, epoch=k, ld=ld, out=(ids[:G], ln.view(-1)[:G]), pad=rows_pad, u16=rows_u16, **kw)

    def launch(j, ln, count=E, first=None, pad=None, u16=None, out_ids=None, packed=None, slab=True):
        """epochs first .. first + count - 1 (default: launch j of the timed region) in one gtok_sent call"""
        o = ids if out_ids is None else out_ids
        return gtok.ops.sent(batch, max_nodes, max_len, seed=0, epoch=j * E if first is None else first, ld=ld,
                             out=(o[:count * G], ln.view(-1)[:count * G]) if slab else None, pad=rows_pad if pad is None else pad, epochs=count,
                             u16=(rows_u16 if out_ids is None else True) if u16 is None else u16, packed=packed, slab=slab, **kw)

    def timed_launches(first_epoch):
        """the K steps: n_launch calls, epochs first_epoch .. first_epoch + K - 1, lengths of step k in lens_all[k]"""
        def f(j):
            cnt = min(E, args.steps - j * E)
            launch(j, lens_all[j * E:j * E + cnt], count=cnt, first=first_epoch + j * E)
        return f

    # the requested protocol first, from a cold start (the device idled through the CSR build): W warm-up steps, then the K
    # steps - reported as `cold_start`; the headline below is measured the same way AFTER the >= 1 s leg (working clocks)
    for w in range(-(-args.warmup // E)):
        launch(w, scratch_len)
    cold_wall, cold_ms = timed_loop(timed_launches(args.warmup), n_launch, multi, per_launch_events=False)
    cold = dict(ms_per_step=round(cold_wall / args.steps * 1e3, 4), kernel_ms_per_step=round(float(np.sum(cold_ms)) / args.steps, 4),
                graphs_per_sec=round(world * G * args.steps / cold_wall, 1),
                note="the same W warm-up + K timed steps right after the CSR build, before the sustained leg (the r01 / r02 protocol)")
    # steady state: >= 1 s of back-to-back launches (clocks, L2 / MALL state of a slab rewritten in place), one event pair
    # around all of them - reported as `sustained`.  It runs BEFORE the K timed steps of the headline, not after them: a device
    # that has been idle through the CSR build takes longer than W + K launches (~2 ms) to reach its working clocks, and the
    # K-step figure is meant to be the rate an epoch loop sees
    sustained = None
    if not args.no_sustained:
        for w in range(3):
            launch(w, scratch_len)
        _, est = timed_loop(lambda k: launch(k, scratch_len), 5, multi, per_launch_events=False)
        n_s = max(n_launch, min(200000, int(1.2 / max(float(est[0]) * 1e-3, 2e-5))))
        n_s = int(gtok.dist.all_reduce_max_int(n_s, dev)) if multi else n_s
        _, sm = timed_loop(lambda k: launch(k, scratch_len), n_s, multi, per_launch_events=False)
        sustained = dict(launches=n_s, epochs_per_launch=E, seconds=round(float(sm[0]) * n_s * 1e-3, 3), ms_per_step=round(float(sm[0]) / E, 5),
                         graphs_per_sec=round(G * E / float(sm[0]) * 1e3, 1), order="before the K timed steps")
        log(f"[bench] sustained leg done: {float(sm[0]) / E:.5f} ms per step over {n_s} launches of {E} epochs")
    for w in range(-(-args.warmup // E)):
        launch(w, scratch_len)
    # the timed region: exactly K steps (ceil(K / E) launches), barrier + synchronize on both sides, one HIP event pair around
    # the launches (roofline.kernel_ms = elapsed / launches: launch-to-launch, the ~1.5 us kernel boundary included)
    wall, kern_ms = timed_loop(timed_launches(args.warmup), n_launch, multi, per_launch_events=False)
    # cross-check outside the timed region: the same launches with an event pair around every one (the kernel alone;
    # what rocprofv3 --kernel-trace --stats reports for it)
    _, kern_each_ms = timed_loop(timed_launches(args.warmup), n_launch, multi)
    tmax = torch.tensor([wall], dtype=torch.float64, device=dev)
    if multi:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall = float(tmax.item())
    # beside the headline: the same steps with GTOK_SENT_NO_PAD (rows written up to their length only - what
    # TokenizedGraphDataset uses, its readers go through `len`); not the headline because the C ABI's documented output is the padded slab
    nopad_ms = ragged = u16_ms = u16p_ms = ragged16 = ragged16_two = ragged16_scan = ragged16_slab = epoch_loop = packed_flag = None
    if zinc and not args.no_unpadded and not rows_u16:
        per_step = lambda ms: float(np.sum(ms)) / (n_launch * E)
        _, npm = timed_loop(lambda j: launch(j, scratch_len, pad=False), n_launch, multi, per_launch_events=False)
        nopad_ms = per_step(npm)
        # 16-bit rows (GTOK_SENT_U16): the walk kernels hold tokens 16 bits each, the rows leave as they are - what
        # TokenizedGraphDataset, EpochRows, gtok_collate_packed (row_ptr NULL) and the compact all-gather read
        ids16 = torch.empty((E * G, ld), dtype=torch.int16, device=dev)
        launch(0, scratch_len, pad=False, u16=True, out_ids=ids16)
        _, um = timed_loop(lambda j: launch(j, scratch_len, pad=False, u16=True, out_ids=ids16), n_launch, multi, per_launch_events=False)
        u16_ms = per_step(um)
        _, upm = timed_loop(lambda j: launch(j, scratch_len, pad=True, u16=True, out_ids=ids16), n_launch, multi, per_launch_events=False)
        u16p_ms = per_step(upm)
        # ragged rows: tokens only (no pad tails) + one pass that packs them back to back (row_ptr + ids) - the form the compact
        # all-gather and the D2H copy of an epoch move; from the int32 slab (round 3) and from the 16-bit slab
        def ragged_step(k):
            launch(k, scratch_len, pad=False)
            ptr = gtok.ops.row_offsets(scratch_len, ld)
            return gtok.ops.pack_rows(ids, scratch_len, ptr, elem_bytes=4, capacity=E * G * ld, check_status=False)
        ragged_step(0)
        _, rgm = timed_loop(ragged_step, n_launch, multi, per_launch_events=False)
        ragged = per_step(rgm)

        def ragged16_scan_step(k):   # offsets + packing in one pass (gtok_pack_rows_scan: the capacity is known before the sizes are)
            launch(k, scratch_len, pad=False, u16=True, out_ids=ids16)
            return gtok.ops.pack_rows_u16(ids16, scratch_len, None, elem_bytes=2, capacity=E * G * ld, check_status=False)
        ragged16_scan_step(0)
        _, rgm = timed_loop(ragged16_scan_step, n_launch, multi, per_launch_events=False)
        ragged16_scan = per_step(rgm)

        # the walk packs its rows itself (gtok_sent_packed, ABI v6): no second pass; the buffer is sized from the first launch + 4 %
        launch(0, scratch_len, pad=False, u16=True, out_ids=ids16)
        need = int(((scratch_len.view(-1)[:E * G].clamp(0, ld) + 7) // 8 * 8).sum().item())
        pk = gtok.ops.PackedRows(E * G, int(need * 1.04) + 4096, True, dev)

        def ragged16_slab_step(k):   # ... beside the 16-bit slab
            launch(k, scratch_len, pad=False, u16=True, out_ids=ids16, packed=pk)
        ragged16_slab_step(0)
        packed_flag = None if pk.fused and not int(pk.status().item()) else f"beside the slab: fused {pk.fused}, status {int(pk.status().item())}"
        _, rgm = timed_loop(ragged16_slab_step, n_launch, multi, per_launch_events=False)
        ragged16_slab = per_step(rgm)

        def ragged16_step(k):        # ... alone (GTOK_SENT_PACK_ONLY: no slab, the rows are staged in 64 rows per resident wave)
            launch(k, scratch_len, u16=True, packed=pk, slab=False)
        ragged16_step(0)
        if not pk.fused or int(pk.status().item()):       # (reported in the line, never raised: every rank must reach the timed loops below)
            packed_flag = f"alone: fused {pk.fused}, status {int(pk.status().item())}"
        _, rgm = timed_loop(ragged16_step, n_launch, multi, per_launch_events=False)
        ragged16 = per_step(rgm)
        del pk

        def ragged16_two_step(k):    # round 4's route: gtok_row_offsets (three launches) + gtok_pack_rows_u16
            launch(k, scratch_len, pad=False, u16=True, out_ids=ids16)
            ptr = gtok.ops.row_offsets(scratch_len, ld)
            return gtok.ops.pack_rows_u16(ids16, scratch_len, ptr, elem_bytes=2, capacity=E * G * ld, check_status=False)
        ragged16_two_step(0)
        _, rgm2 = timed_loop(ragged16_two_step, n_launch, multi, per_launch_events=False)
        ragged16_two = per_step(rgm2)
        del ids16
    # what an epoch of the dataset classes costs on the device: Graph2TrailTokenizer.epochs_for(G) epochs per launch as 16-bit
    # rows without padding (agtt.TokenizedGraphDataset.tokenize_epoch_u16 / the tokenizer's _serve) - every workload
    if not args.no_unpadded and not rows_u16:
        Kd = gtok.Graph2TrailTokenizer.epochs_for_shape(G, ld)
        idsk = torch.empty((Kd * G, ld), dtype=torch.int16, device=dev)
        lnk = torch.empty((Kd * G,), dtype=torch.int32, device=dev)
        fk = lambda j: gtok.ops.sent(batch, max_nodes, max_len, seed=0, epoch=j * Kd, ld=ld, out=(idsk, lnk), pad=False, epochs=Kd, u16=True, **kw)
        for _ in range(3):
            fk(0)
        nk = max(10, -(-args.steps // Kd))      # (round 4 timed 3 launches: 0.8656 ms per launch in the driver's run against 0.8188 in rocprof's 1,418)
        _, km = timed_loop(fk, nk, multi, per_launch_events=False)
        tok_k = float(lnk.to(torch.float64).sum().item())           # tokens of the last launch (Kd epochs)
        bytes_k = host.algorithmic_read_bytes(ibtt=False, labeled=zinc) * Kd + 2.0 * tok_k + 4.0 * G * Kd
        epoch_loop = dict(epochs_per_launch=Kd, launches=nk, ms_per_launch=round(float(np.mean(km)), 4), ms_per_epoch=round(float(np.mean(km)) / Kd, 5),
                          graphs_per_sec=round(G * Kd / float(np.mean(km)) * 1e3, 1),
                          roofline=dict(bound="hbm", limiter="valu_issue", achieved=round(bytes_k / (float(np.mean(km)) * 1e-3) / 1e9, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                                        frac=round(bytes_k / (float(np.mean(km)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), algorithmic_bytes_per_launch=int(bytes_k),
                                        note="SURVEY section 8d with 16-bit ids: 4(N+1) + 4E (+ N + E labelled) read, 2L + 4 written, x epochs per launch"),
                          kernel=gtok.ops.sent_kernel_name(batch, max_nodes, max_len, labeled=zinc, num_node_types=ntypes, num_edge_types=etypes, remap_zinc=zinc, epochs=Kd),
                          note="the dataset classes' epoch: tokenizer.epochs_for(G) epochs per gtok_sent launch, GTOK_SENT_U16 | GTOK_SENT_NO_PAD rows "
                               "read in place by gtok_collate_packed / EpochRows")
        del idsk, lnk
    log(f"[bench] timed region done: {wall / args.steps * 1e3:.5f} ms per step ({n_launch} launches of {E} epochs)")
    # one-off layout steps ops.sent did inside the warm-up (like the CSR build: once per resident batch, never per epoch):
    # their time and bytes, re-measured on a fresh copy of the batch, and the kernel WITHOUT any of them (what a C-ABI
    # caller that passes only the int32 CSR gets)
    layout = {}

    def fresh():
        b = gtok.GraphBatch(batch.num_graphs, batch.max_nodes, batch.max_edges, batch.node_ptr, batch.edge_ptr, batch.rowptr, batch.col,
                            batch.eorder, batch.nattr, batch.eattr, batch.flags, batch.chunk_nodes, batch.chunk_edges, batch.max_degree)
        return b

    def once(f):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3, r
    nbytes = lambda *ts: int(sum(t.numel() * t.element_size() for t in ts if t is not None))
    if batch.rowptr8 is not None or (batch.lane_sorted is not None and batch.lane_sorted.rowptr8 is not None):
        fb = fresh(); once(lambda: gtok.ops.pack8(fresh()))
        ms, _ = once(lambda: gtok.ops.pack8(fb))
        layout["pack8"] = dict(ms=round(ms, 3), bytes=nbytes(fb.rowptr8, fb.col8))
    if batch.lane_sorted is not None:
        fb = fresh()
        ms, sb = once(lambda: gtok.ops.lane_sorted(fb))
        layout["lane_sorted"] = dict(ms=round(ms, 3), units=sb.num_units,
                                     bytes=nbytes(sb.node_ptr, sb.edge_ptr, sb.rowptr, sb.col, sb.nattr, sb.eattr, sb.rowptr8, sb.col8, sb.graph_ids, sb.unit_ptr))
    if batch.adj_rows is not None:
        fb = fresh()
        ms, _ = once(lambda: gtok.ops.adjbits(fb))
        layout["adjbits"] = dict(ms=round(ms, 3), bytes=nbytes(fb.adj_rows, fb.adj_planes))
        ms, lo = once(lambda: gtok.ops._lane_order(fb, max(1, max_len)))
        layout["lane_order"] = dict(ms=round(ms, 3), bytes=nbytes(lo))
    if layout and not rows_u16:
        saved = {k: os.environ.get(k) for k in ("GTOK_NO_LANE_SORT", "GTOK_NO_PACK8", "GTOK_NO_ADJBITS")}
        os.environ.update(GTOK_NO_LANE_SORT="1", GTOK_NO_PACK8="1", GTOK_NO_ADJBITS="1")
        try:
            fb = fresh()
            one = (ids[:G], scratch_len[:G])
            for w in range(args.warmup):
                gtok.ops.sent(fb, max_nodes, max_len, seed=0, epoch=w, ld=ld, out=one, **kw)
            _, nm = timed_loop(lambda k: gtok.ops.sent(fb, max_nodes, max_len, seed=0, epoch=args.warmup + k, ld=ld, out=one, **kw),
                               args.steps, multi, per_launch_events=False)
            layout["without_any_mirror"] = dict(kernel=gtok.ops.sent_kernel_name(fb, max_nodes, max_len, labeled=zinc, num_node_types=ntypes,
                                                                                 num_edge_types=etypes, remap_zinc=zinc),
                                                ms_per_step=round(float(np.mean(nm)), 4), graphs_per_sec=round(G / float(np.mean(nm)) * 1e3, 1),
                                                note="the same batch as plain int32 CSR (no byte mirror, no reordered copy, no bit matrix)")
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        step(args.warmup, scratch_len)      # back on the resident layouts for the legs below

    # the same K steps through the PyTorch custom op (north_star: "exposed as PyTorch-ROCm custom ops"): torch.ops.gtok.sent on
    # the arrays torch.ops.gtok.csr_prepare hands back (prepared once, like the CSR build), and on the raw tensors alone (the op
    # prepares behind the call on first use and finds the batch again by tensor identity) - same launches, same output buffers
    torch_op = None
    if zinc and E == 1 and not rows_u16 and rows_pad:
        try:
            raw = dict(node_ptr=batch.node_ptr, edge_ptr=batch.edge_ptr, rowptr=batch.rowptr, col=batch.col, nattr=batch.nattr, eattr=batch.eattr)
            t0p = time.perf_counter()
            P = gtok.torch_ops.prepared_args(**raw, max_nodes=batch.max_nodes, max_edges=batch.max_edges)
            torch.cuda.synchronize()
            prep_ms = (time.perf_counter() - t0p) * 1e3
            t0p = time.perf_counter()
            P = gtok.torch_ops.prepared_args(**raw, max_nodes=batch.max_nodes, max_edges=batch.max_edges)
            torch.cuda.synchronize()
            prep_ms = min(prep_ms, (time.perf_counter() - t0p) * 1e3)
            opkw = dict(query=None, max_num_nodes=max_nodes, max_len=max_len, ld=ld, seed=0, labeled=True, num_node_types=ntypes,
                        num_edge_types=etypes, remap_zinc=True, pad_id=5, graph_base=graph_base)
            res = {}
            for label, arrays in (("prepared", P), ("raw_tensors", dict(raw, max_nodes=batch.max_nodes, max_edges=batch.max_edges))):
                f = lambda k: torch.ops.gtok.sent(**arrays, epoch=args.warmup + k, **opkw)
                # the headline's protocol: a leg of back-to-back launches first (working clocks: the legs before this one ran other
                # kernels in short bursts - measured right behind them the same launches read 7 % slower), then W + K
                if sustained is not None:
                    timed_loop(f, max(args.steps, sustained["launches"] // 4), multi, per_launch_events=False)
                for w in range(args.warmup):
                    f(w)
                kname_op = gtok.ops.last_sent_kernel()
                _, om = timed_loop(f, args.steps, multi, per_launch_events=False)
                res[label] = dict(ms_per_step=round(float(np.mean(om)), 5), kernel=kname_op)
            o_ids, o_ln = torch.ops.gtok.sent(**P, epoch=args.warmup + args.steps - 1, **opkw)
            step(args.warmup + args.steps - 1, scratch_len)
            same = bool(torch.equal(o_ids, ids[:G]) and torch.equal(o_ln, scratch_len[:G]))
            torch_op = dict(ms_per_step=res["prepared"]["ms_per_step"], kernel=res["prepared"]["kernel"], graphs_per_sec=round(G / res["prepared"]["ms_per_step"] * 1e3, 1),
                            raw_tensors_ms_per_step=res["raw_tensors"]["ms_per_step"], raw_tensors_kernel=res["raw_tensors"]["kernel"],
                            csr_prepare_ms=round(prep_ms, 3), equals_ops_sent=same,
                            note="torch.ops.gtok.sent over K steps, one HIP event pair around them; output allocated by the op each call (torch's "
                                 "caching allocator); prepared = torch_ops.prepared_args (gtok_csr_lane_sort on the device) once, raw_tensors = "
                                 "the op prepares behind the first call and finds the batch again by tensor identity")
            del o_ids, o_ln, P
        except Exception as ex:      # a secondary leg must never cost the run its headline line
            import traceback
            traceback.print_exc()
            torch_op = {"error": f"{type(ex).__name__}: {ex}"}

    all_len = lens_all[:args.steps]
    if int(all_len.max().item()) > ld:
        raise SystemExit(f"slab width {ld} too narrow for a timed step (max len {int(all_len.max())}): rerun with --ld safe")
    tok_total = torch.tensor([float(all_len.sum().item())], dtype=torch.float64, device=dev)
    if multi:
        dist.all_reduce(tok_total)
    tokens_per_step_rank = float(all_len.sum().item()) / args.steps
    value = world * G * args.steps / wall
    tokens_per_sec = float(tok_total.item()) / wall

    # roofline of the dominant kernel (sent_kernel<1,true>): algorithmic bytes / launch duration
    read_b = host.algorithmic_read_bytes(ibtt=False, labeled=zinc)
    id_bytes = 2.0 if rows_u16 else 4.0               # (--rows u16 | u16padded, the counter-pass flavours: rows of 16-bit ids)
    write_b = id_bytes * tokens_per_step_rank + 4.0 * G
    # per LAUNCH: the figure of SURVEY section 8d per graph x the graphs one launch processes (E epochs of the corpus: every
    # (unit, epoch) pair stages its CSR chunk and writes its rows) over the average launch duration
    epl = args.steps / n_launch                       # epochs per launch, averaged over the timed region (= E when E divides K)
    read_b, write_b = read_b * epl, write_b * epl
    kern_s = float(np.mean(kern_ms)) * 1e-3
    achieved = (read_b + write_b) / kern_s / 1e9
    kname = gtok.ops.sent_kernel_name(batch, max_nodes, max_len, labeled=zinc, num_node_types=ntypes, num_edge_types=etypes,
                                      remap_zinc=zinc, epochs=E) + ("<labelled>" if zinc else "<unlabelled>")
    # HBM bytes per launch come from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass, and counter
    # collection perturbs the timing), i.e. from an EARLIER run of this same command: profiles/pmc_traffic.json records
    # the kernel and the commit it was measured at, and the figure is dropped when the kernel chosen now differs.
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        rec = json.load(open(tpath)).get(f"sent:{args.workload}:{G}:{args.ld}" + (f":x{E}" if E > 1 else ""))
        if rec and rec.get("kernel_label", kname) == kname:
            traffic = rec["hbm_bytes_per_launch"]
            traffic_source = f"{rec.get('source', 'profiles/pmc_traffic.json')} (rocprofv3 --pmc, measured at commit {rec.get('commit', 'unknown')}, not in this run)"
    # rows cut at max_len stop the walk early: the bytes such a walk NEEDS are those of the rows it visited.  Visited nodes per
    # row come from the tokens (sent_decode); the entries behind them are taken in proportion (E_g * visited / N_g), so this
    # second yardstick is an estimate, reported beside the fixed one of SURVEY section 8d
    trunc = None
    if not zinc and not rows_u16:
        step(args.warmup, scratch_len)
        # (count-only decode: capacities of 0 - the row is read to its end; round 3 passed capacities of 4 to a decoder that
        # stopped at the first entry that did not fit and reported ~5 nodes per walk where the walks reach ~68)
        dec = gtok.ops.sent_decode(ids[:G], scratch_len[:G], max_nodes, False, 0, edge_cap=0, node_cap=0)
        vis_n = dec["num_nodes"].to(torch.float64)
        nc_t = (batch.node_ptr[1:] - batch.node_ptr[:-1]).to(torch.float64).clamp(min=1)
        ec_t = (batch.edge_ptr[1:] - batch.edge_ptr[:-1]).to(torch.float64)
        need_read = float((4 * (vis_n + 1) + 4 * ec_t * (vis_n / nc_t).clamp(max=1.0)).sum().item()) * epl
        tb = need_read + write_b
        trunc = dict(bytes_per_launch=int(tb), achieved=round(tb / kern_s / 1e9, 2), frac=round(tb / kern_s / 1e9 / HBM_PEAK_GBS, 5),
                     rows_cut=int((scratch_len[:G] >= max_len).sum().item()), avg_nodes_visited=round(float(vis_n.mean().item()), 2),
                     note="4(k+1) + 4 E k/N + 4L + 4 with k = nodes the walk reached before max_len (estimate: entries in proportion)")
    # `bound` names the roofline the fraction is taken against (HBM bytes).  What the kernels actually run into is
    # the vector issue rate - counters in profiles/r0*/pmc_summary_*: the molecule kernel issues one vector instruction per ~3.9
    # SIMD-cycles at 16 epochs per launch (saturated), the bit-matrix kernel keeps its SIMDs' vector pipes ~70 % busy with two
    # waves each - so the line says so instead of letting the HBM fraction read as a bandwidth problem.
    roofline = dict(bound="hbm", limiter="valu_issue", kernel=kname, achieved=round(achieved, 2), peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=traffic_source,
                    algorithmic_bytes_per_launch=int(read_b + write_b), kernel_ms=round(kern_s * 1e3, 4),
                    kernel_ms_event_pair_per_launch=round(float(np.mean(kern_each_ms)), 4),
                    epochs_per_launch=E, launches=n_launch, kernel_ms_per_epoch=round(kern_s * 1e3 / epl, 5),
                    padded_slab_bytes_per_launch=int(id_bytes * G * ld * epl))
    if trunc is not None:
        roofline["truncation_aware"] = trunc
        if epoch_loop is not None:       # the same yardstick for the dataset classes' K-epoch launches (16-bit rows: 2 bytes per token)
            ek = epoch_loop["epochs_per_launch"]
            tbk = (need_read / epl + 2.0 * tokens_per_step_rank + 4.0 * G) * ek
            ks = epoch_loop["ms_per_launch"] * 1e-3
            epoch_loop["roofline"]["truncation_aware"] = dict(bytes_per_launch=int(tbk), achieved=round(tbk / ks / 1e9, 2),
                                                              frac=round(tbk / ks / 1e9 / HBM_PEAK_GBS, 5),
                                                              note="rows visited before max_len only (as roofline.truncation_aware), 2L + 4 written")

    out = dict(metric="graphs_tokenized_per_sec", value=round(value, 1), unit="graphs/s", n_gpus=world,
               steps=args.steps, warmup=args.warmup, ms_per_step=round(wall / args.steps * 1e3, 5),
               higher_is_better=True, scaling="weak", vs_baseline=None, dtype="u16" if rows_u16 else "int32", data="synthetic",
               tokens_per_sec=round(tokens_per_sec, 1),
               config=dict(workload=f"{args.workload}: {wl['desc']}" + ("" if args.rows == "padded" else f" [--rows {args.rows}]"), graphs_per_gpu=G, max_len=max_len,
                           slab_width=ld, slab_width_mode=args.ld, avg_tokens_per_graph=round(tokens_per_step_rank / G, 2),
                           parallelism=f"graph-sharded x{world}, no data-path collective",
                           epochs_per_launch=E, launches_in_timed_region=n_launch,
                           methodology="a step = one epoch pass; K steps = ceil(K / E) gtok_sent launches of E epochs; since round 3 the >= 1 s "
                                       "sustained leg runs BEFORE the K timed steps (working clocks) - `cold_start` is the same W + K protocol "
                                       "right after the CSR build, comparable with the r01 / r02 records"),
               roofline=roofline, cold_start=cold)
    # layout steps done once per resident batch by ops.sent (outside the timed region, like the CSR build): say which were used
    layouts = [n for n, on in (("byte mirror of rowptr/col", batch.rowptr8 is not None or (batch.lane_sorted is not None and batch.lane_sorted.rowptr8 is not None)),
                               ("copy reordered by expected walk length (graph_ids + unit table)", batch.lane_sorted is not None),
                               ("adjacency bit-matrix mirror + lane order", batch.adj_rows is not None)) if on]
    out["config"]["resident_layouts"] = layouts
    if sustained is not None:
        out["sustained"] = sustained
    if layout:
        out["layout"] = layout
    if nopad_ms is not None:
        out["unpadded_rows"] = dict(ms_per_step=round(nopad_ms, 4), graphs_per_sec=round(G / nopad_ms * 1e3, 1),
                                    note="GTOK_SENT_NO_PAD: tokens only, pad tails of the slab not written")
        out["ragged_rows"] = dict(ms_per_step=round(ragged, 4), graphs_per_sec=round(G / ragged * 1e3, 1),
                                  note="GTOK_SENT_NO_PAD + gtok_row_offsets + gtok_pack_rows (int32): rows back to back behind row_ptr "
                                       "(the round-3 route to gtok_collate_packed / the compact exchange)")
        out["u16_rows"] = dict(unpadded_ms_per_step=round(u16_ms, 4), unpadded_graphs_per_sec=round(G / u16_ms * 1e3, 1),
                               padded_ms_per_step=round(u16p_ms, 4), padded_graphs_per_sec=round(G / u16p_ms * 1e3, 1),
                               packed_ms_per_step=round(ragged16, 4), packed_beside_slab_ms_per_step=round(ragged16_slab, 4),
                               packed_scan_ms_per_step=round(ragged16_scan, 4),
                               packed_two_pass_ms_per_step=round(ragged16_two, 4),
                               **({"packed_legs_not_as_described": packed_flag} if packed_flag else {}),
                               note="GTOK_SENT_U16: rows of 16-bit ids straight from the walk's token windows (no unpacking, half the bytes); "
                                    "unpadded = + GTOK_SENT_NO_PAD: what TokenizedGraphDataset / gtok_collate_packed(row_ptr NULL) / EpochRows read in place, "
                                    "no second pass; packed = gtok_sent_packed + GTOK_SENT_PACK_ONLY: the walk appends every finished unit's rows to a packed buffer (row starts "
                                    "beside the lengths: the compact all-gather's payload, no second pass, no slab - rows are staged in 64 rows per resident wave; "
                                    "zeroing the 8 KB of fill marks is inside the figure); packed_beside_slab = gtok_sent_packed writing the unpadded 16-bit slab as well; "
                                    "packed_scan = unpadded + gtok_pack_rows_scan (offsets + rows back to back in dataset order in one pass); "
                                    "packed_two_pass = + gtok_row_offsets + gtok_pack_rows_u16 (round 4's route, four launches)")
    if epoch_loop is not None:
        out["epoch_loop"] = epoch_loop

    log("[bench] headline assembled; secondary legs follow")
    # IBTT serialiser on the same corpus (second half of the metric; outside the timed region)
    if not args.no_ibtt:
        if zinc:
            vocab = {t: i for i, t in enumerate(
                ["<bos>", "<eos>", "<pad>", "<unk>", "<q>", "<p>", "<atom>", "<bond>", "C", "N", "O", "F", "P", "S", "Cl",
                 "Br", "I", "single", "double", "triple", "aromatic", "regression"]
                + [str(i) for i in range(max_nodes)] + ["X", "unknown"])}
            lut = gtok.ops.zinc_lut(vocab, max_nodes).to(dev)
            _, l0 = gtok.ops.ibtt_zinc(batch, lut, max_len, vocab["<pad>"])
            ild = (int(l0.max().item()) + 3) // 4 * 4                      # lengths are deterministic: exact width
            run = lambda out: gtok.ops.ibtt_zinc(batch, lut, max_len, vocab["<pad>"], ld=ild, out=out)
            ibtt_read = host.algorithmic_read_bytes(ibtt=True, labeled=True)
        else:
            vocab = {t: i for i, t in enumerate(["<pad>", "<bos>", "<e>", "<n>", "<q>", "<p>", "<eos>", "yes", "no", "has_cycle"]
                                                + [str(i) for i in range(max_nodes)])}
            lut = gtok.ops.synth_lut(vocab, max_nodes).to(dev)
            q = torch.zeros((G, 4), dtype=torch.int32, device=dev); q[:, 0] = 1; q[:, 1] = vocab["has_cycle"]
            _, l0 = gtok.ops.ibtt_synth(batch, lut, q, max_len, vocab["<pad>"])
            ild = (int(l0.max().item()) + 3) // 4 * 4
            run = lambda out: gtok.ops.ibtt_synth(batch, lut, q, max_len, vocab["<pad>"], ld=ild, out=out)
            # the serialiser stops reading edges once max_len tokens are out: 4(N+1) rowptr + 4*min(E, max_len/3) col
            ecap = torch.clamp(torch.from_numpy(d["edge_counts"]), max=(max_len + 2) // 3)
            ibtt_read = 4 * (host.num_nodes_total + G) + 4 * int(ecap.sum())
        iids = torch.empty((G, ild), dtype=torch.int32, device=dev)
        iln = torch.empty((G,), dtype=torch.int32, device=dev)
        f = lambda k: run((iids, iln))
        for _ in range(args.warmup):
            f(0)
        iwall, ik_ms = timed_loop(f, args.steps, multi)
        itok = float(iln.sum().item())
        ib = ibtt_read + 4.0 * itok + 4.0 * G
        ik = float(np.mean(ik_ms)) * 1e-3
        if zinc:
            import ctypes
            _cs = batch.c_struct()
            iname = gtok.lib().gtok_ibtt_zinc_kernel_name(ctypes.byref(_cs)).decode()
        else:
            iname = "ibtt_synth_kernel"
        itraffic, itraffic_source = None, None
        if os.path.exists(tpath):
            rec = json.load(open(tpath)).get(f"ibtt:{args.workload}:{G}")
            if rec and rec.get("kernel_label", iname) == iname:
                itraffic = rec["hbm_bytes_per_launch"]
                itraffic_source = f"{rec.get('source', 'profiles/pmc_traffic.json')} (rocprofv3 --pmc, measured at commit {rec.get('commit', 'unknown')}, not in this run)"
        out["ibtt"] = dict(kernel=iname,
                           graphs_per_sec_per_gpu=round(G * args.steps / iwall, 1),
                           tokens_per_sec_per_gpu=round(itok * args.steps / iwall, 1), kernel_ms=round(ik * 1e3, 4),
                           slab_width=ild, avg_tokens_per_graph=round(itok / G, 2),
                           roofline=dict(bound="hbm", achieved=round(ib / ik / 1e9, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                                         frac=round(ib / ik / 1e9 / HBM_PEAK_GBS, 5), traffic=itraffic, traffic_source=itraffic_source,
                                         algorithmic_bytes_per_launch=int(ib),
                                         padded_slab_bytes_per_launch=int(4 * G * ild)))

    # the corpus pass of build_vocab_from_texts, from the CSR (graph-token corpora; outside the timed region)
    if not zinc and not args.no_ibtt:
        nid = max_nodes
        if nid <= 1024:
            vc = torch.zeros(nid, dtype=torch.int64, device=dev)
            vf = torch.full((nid,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
            fv = lambda k: gtok.ops.vocab_stats_synth(batch, nid, out=(vc, vf))
            for _ in range(args.warmup):
                fv(0)
            vwall, vk_ms = timed_loop(fv, args.steps, multi)
            vb = 4.0 * (host.num_nodes_total + G) + 4.0 * host.num_edges_total
            vk = float(np.mean(vk_ms)) * 1e-3
            out["vocab_stats"] = dict(kernel="vocab_stats_synth_kernel", graphs_per_sec_per_gpu=round(G * args.steps / vwall, 1),
                                      kernel_ms=round(vk * 1e3, 4),
                                      roofline=dict(bound="hbm", achieved=round(vb / vk / 1e9, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                                                    frac=round(vb / vk / 1e9 / HBM_PEAK_GBS, 5), traffic=None,
                                                    algorithmic_bytes_per_launch=int(vb)))

    # TokenDataset's text -> ids on the graph-token TEXT of a bounded slice of the same corpus (the
    # texts are rendered on the host, which is the slow part; outside the timed region)
    if not zinc and not args.no_ibtt:
        S = min(G, 4096)
        nc, ec = d["node_counts"][:S], d["edge_counts"][:S]
        eptr = np.concatenate([[0], np.cumsum(ec)])
        texts = []
        for g in range(S):
            u = d["src"][eptr[g]:eptr[g + 1]].tolist(); v = d["dst"][eptr[g]:eptr[g + 1]].tolist()
            body = " ".join(f"{a} {b} <e>" for a, b in zip(u, v))
            texts.append(" ".join(t for t in ("<bos>", body, "<n>", " ".join(map(str, range(int(nc[g])))),
                                              "<q> has_cycle <p> yes <eos>") if t))
        tb, tp = gtok.ops.pack_texts(texts)
        tb, tp = tb.to(dev), tp.to(dev)
        table = gtok.ops.VocabTable(vocab, dev)
        tids = torch.empty((S, ild), dtype=torch.int32, device=dev); tln = torch.empty(S, dtype=torch.int32, device=dev)
        ft = lambda k: gtok.ops.text_to_ids(tb, tp, table, max_len, True, ld=ild, out=(tids, tln))
        for _ in range(args.warmup):
            ft(0)
        twall, tk_ms = timed_loop(ft, args.steps, multi)
        same = bool(torch.equal(tids, iids[:S]) and torch.equal(tln, iln[:S]))     # the CSR path emits the same ids
        tk = float(np.mean(tk_ms)) * 1e-3
        # the tokenizer stops at max_len tokens: algorithmic read = the text up to the end of the last kept token
        need = sum(len(" ".join(t.split()[:max_len])) for t in texts)
        tbytes = float(need) + 4.0 * float(tln.sum().item()) + 4.0 * S
        out["text_to_ids"] = dict(kernel="text_ids_kernel", texts=S, text_bytes=int(tb.numel()), text_bytes_needed=int(need),
                                  graphs_per_sec_per_gpu=round(S * args.steps / twall, 1), kernel_ms=round(tk * 1e3, 4),
                                  equals_csr_path=same,
                                  roofline=dict(bound="hbm", achieved=round(tbytes / tk / 1e9, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                                                frac=round(tbytes / tk / 1e9 / HBM_PEAK_GBS, 5), traffic=None,
                                                algorithmic_bytes_per_launch=int(tbytes)))

    # The two exchange legs run last among the multi-rank work and must never cost the run its headline line: an error that every
    # rank hits alike (a Python-level failure) is recorded in the line instead of raised.  (A rank that fails ALONE inside a
    # collective still ends the run - the others wait until the process group's timeout.)
    def exchange_legs():
        # reassembling the padded slab on every rank: one RCCL all-gather over xGMI, timed on its own
        if multi:
            log("[bench] all-gather legs")
            gtok.dist.gather_tokens(ids[:G], lens[-1], world * G, 5, force=True)
            torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                gtok.dist.gather_tokens(ids[:G], lens[-1], world * G, 5, force=True)
            torch.cuda.synchronize(); dist.barrier()
            ag = (time.perf_counter() - t0) / reps
            agt = torch.tensor([ag], dtype=torch.float64, device=dev)
            dist.all_reduce(agt, op=dist.ReduceOp.MAX)
            recv = (world - 1) * G * (ld + 1) * 4
            out["allgather"] = dict(ms=round(float(agt.item()) * 1e3, 3), bytes_received_per_gpu=recv,
                                    bytes_gathered_per_gpu=world * G * (ld + 1) * 4,
                                    GBps_per_gpu=round(world * G * (ld + 1) * 4 / float(agt.item()) / 1e9, 2))

        # BASELINE config 4 as configured: ONE ZINC-full corpus block-sharded over the ranks (strong scaling), each rank
        # tokenizes its block with graph_base = the block's first global index, one all-gather reassembles the padded slab
        # on every rank.  Reported beside the weak-scaling line above (the driver computes efficiency from `value`).
        if multi and zinc and (args.scaling == "strong" or os.environ.get("GTOK_BENCH_STRONG", "1") == "1"):
            Gt = args.graphs or wl["graphs"]
            dc = gtok.synth.zinc_like(Gt, seed=1000)                      # the same corpus on every rank
            whole = gtok.GraphBatch.from_coo_device(dc["node_counts"], dc["edge_counts"], dc["src"], dc["dst"], dc["x"], dc["edge_attr"], device=dev)
            mine, lo, hi = gtok.dist.shard_of(whole, rank, world)
            del whole
            gtok.ops.pack8(mine)
            kws = dict(kw, graph_base=lo)
            # K epochs per launch (round 4): a rank's 31 k-molecule share cannot fill the chip for one epoch (0.065 ms against
            # 0.076 ms for the whole corpus on one GPU), Es epochs of it can - and ONE exchange then carries Es epochs of rows.
            # Rows leave the walk as 16-bit ids without padding; the compact exchange packs them as they are.  Each rank's
            # buffer is [Es, per, ld] (epoch-major); the gathered slab is rank-major: epoch e of graph g sits at row
            # ((g // per) * Es + e) * per + g % per - consumers index it (gtok_collate takes any row list).
            Gs, per = hi - lo, -(-Gt // world)
            Es = gtok.Graph2TrailTokenizer.epochs_for_shape(Gs, ld) if Gt % world == 0 else 1
            if os.environ.get("GTOK_BENCH_STRONG_EPOCHS"):
                Es = max(1, int(os.environ["GTOK_BENCH_STRONG_EPOCHS"]))
            sids = torch.empty((Es * Gs, ld), dtype=torch.int16, device=dev)
            sln = torch.empty((Es * Gs,), dtype=torch.int32, device=dev)
            gstats = {}
            spk = {}

            def sstep(k, gather):
                if gather in ("compact", "compact_packed"):
                    # the walk packs the rows itself and writes no slab (gtok_sent_packed, GTOK_SENT_PACK_ONLY); capacity = the first launch's largest rank + 4 %;
                    # compact_packed: the gathered rows are not re-padded either (absolute row starts: what the collate kernels read in place)
                    _, cln = gtok.ops.sent(mine, max_nodes, max_len, seed=0, epoch=k * Es, ld=ld, epochs=Es, u16=True, packed=spk["pk"], slab=False, **kws)
                    return gtok.dist.gather_tokens(None, cln.reshape(-1), Gt * Es, 5, force=True, compact=True, packed=spk["pk"], ld=ld,
                                                   as_packed=gather == "compact_packed", stats=gstats.setdefault(gather, {}))
                gtok.ops.sent(mine, max_nodes, max_len, seed=0, epoch=k * Es, ld=ld, out=(sids, sln), pad=False, epochs=Es, u16=True, **kws)
                if gather == "padded":      # the 16-bit slab as it is (pad tails and all: they are not written, the bytes still travel)
                    return gtok.dist.gather_tokens(sids, sln, Gt * Es, 5, force=True, stats=gstats.setdefault("padded", {}))
                if gather == "compact_scan":    # round 5's first route: pack in a second pass (gtok_pack_rows_scan), capacity as below
                    return gtok.dist.gather_tokens(sids, sln, Gt * Es, 5, force=True, compact=True, capacity=gstats.get("cap"),
                                                   stats=gstats.setdefault("compact_scan", {}))
            res = {}
            n_sl = max(2, -(-args.steps // Es))
            for gather in (None, "padded", "compact_scan", "compact", "compact_packed"):
                if gather == "compact_scan":
                    sstep(0, gather)                              # sized by an all-reduce once ...
                    gstats["cap"] = int(gstats["compact_scan"]["capacity"] * 1.04) + 4096      # ... then a fixed bound, the same on every rank
                    spk["pk"] = gtok.ops.PackedRows(Es * Gs, gstats["cap"], True, dev)
                for w in range(max(1, args.warmup // Es)):
                    sstep(w, gather)
                swall, _ = timed_loop(lambda k: sstep(args.warmup + k, gather), n_sl, True)
                tm = torch.tensor([swall], dtype=torch.float64, device=dev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                res[gather] = float(tm.item())
            if int(sln.max().item()) > ld:
                raise SystemExit("strong-scaling leg: slab too narrow")
            # the compact exchange must give the padded one's rows (checked on the last launch's buffers, inside the row lengths:
            # the 16-bit slab travels without its pad tails written)
            p_ids, p_ln = sstep(args.warmup, "padded")
            c_ids, c_ln = sstep(args.warmup, "compact")
            inside = torch.arange(ld, device=dev)[None, :] < c_ln[:, None]
            same = bool(c_ids.dtype == p_ids.dtype and torch.equal(p_ln, c_ln) and torch.equal(torch.where(inside, p_ids, 0), torch.where(inside, c_ids, 0))
                        and bool((c_ids[~inside] == 5).all())) and int(gstats["compact"]["status"].item()) == 0
            s_ids, s_ln = sstep(args.warmup, "compact_scan")
            same = same and torch.equal(s_ln, c_ln) and torch.equal(s_ids, c_ids) and int(gstats["compact_scan"]["status"].item()) == 0
            (k_buf, k_start), k_ln = sstep(args.warmup, "compact_packed")          # the rows that stayed packed, re-padded here for the comparison only
            k_ids = gtok.ops.unpack_rows_at(k_buf, k_start.contiguous(), k_ln.contiguous(), ld, 5, u16=True)
            same = same and torch.equal(k_ln, c_ln) and torch.equal(k_ids, c_ids) and int(gstats["compact_packed"]["status"].item()) == 0
            epochs_timed = n_sl * Es
            per_epoch = lambda t: round(t / epochs_timed * 1e3, 5)
            out["strong_scaling"] = dict(
                workload=f"one {Gt}-graph corpus block-sharded x{world} (graphs_per_gpu {Gs}), {Es} epochs per launch and per exchange, kernel "
                         f"{gtok.ops.sent_kernel_name(mine, max_nodes, max_len, labeled=True, num_node_types=ntypes, num_edge_types=etypes, remap_zinc=True, epochs=Es)}",
                epochs_per_launch=Es, launches_timed=n_sl,
                tokenize_graphs_per_sec=round(Gt * epochs_timed / res[None], 1), tokenize_ms_per_epoch=per_epoch(res[None]),
                tokenize_and_allgather_graphs_per_sec=round(Gt * epochs_timed / res["compact"], 1),
                tokenize_and_allgather_ms_per_epoch=per_epoch(res["compact"]),
                exchange="compact: 16-bit rows packed by the walk itself, no slab on the sending side (gtok_sent_packed + GTOK_SENT_PACK_ONLY: " + ("fused" if spk["pk"].fused else "NOT fused: gtok_pack_rows_scan behind the walk")
                         + ") + row starts and lengths over RCCL, re-padded locally into a 16-bit slab (gtok_unpack_rows_at; dist.gather_tokens(compact=True, packed=)); "
                         "one exchange per launch = per Es epochs",
                compact=dict(ms_per_epoch=per_epoch(res["compact"]), exchange_ms_per_epoch=per_epoch(res["compact"] - res[None]),
                             bytes_sent_per_rank_per_epoch=gstats["compact"]["bytes_sent_per_rank"] // Es,
                             bytes_gathered_per_rank_per_epoch=world * gstats["compact"]["bytes_sent_per_rank"] // Es),
                compact_packed=dict(ms_per_epoch=per_epoch(res["compact_packed"]), exchange_ms_per_epoch=per_epoch(res["compact_packed"] - res[None]),
                                    bytes_sent_per_rank_per_epoch=gstats["compact_packed"]["bytes_sent_per_rank"] // Es,
                                    note="as compact, and the gathered rows stay packed (dist.gather_tokens(..., as_packed=True) -> (buffer, absolute row starts), lengths): "
                                         "gtok_collate_packed / _batch / _epoch read them in place, nothing is re-padded"),
                compact_scan=dict(ms_per_epoch=per_epoch(res["compact_scan"]), exchange_ms_per_epoch=per_epoch(res["compact_scan"] - res[None]),
                                  bytes_sent_per_rank_per_epoch=gstats["compact_scan"]["bytes_sent_per_rank"] // Es,
                                  note="the rows packed in a second pass (gtok_pack_rows_scan), lengths only beside them"),
                padded=dict(ms_per_epoch=per_epoch(res["padded"]), exchange_ms_per_epoch=per_epoch(res["padded"] - res[None]),
                            bytes_sent_per_rank_per_epoch=gstats["padded"]["bytes_sent_per_rank"] // Es,
                            bytes_gathered_per_rank_per_epoch=world * gstats["padded"]["bytes_sent_per_rank"] // Es,
                            note="the 16-bit slab [Es, per, ld] as it is (half the bytes of round 3's int32 slab)"),
                compact_equals_padded=same, gathered_slab_bytes_per_epoch=int(Gt) * (ld * 2 + 4),
                note="no N > 1 run exists until a SCALE record does: with one rank (GTOK_BENCH_FORCE_DIST=1) the collective is a device-local copy")
            if args.scaling == "strong":      # make the configured workload the headline of this run
                out.update(value=out["strong_scaling"]["tokenize_and_allgather_graphs_per_sec"], scaling="strong",
                           ms_per_step=out["strong_scaling"]["tokenize_and_allgather_ms_per_epoch"])
                out["config"]["parallelism"] = (f"one corpus block-sharded x{world} + compact RCCL all-gather (16-bit rows packed by the walk, row starts, lengths), "
                                                "re-padded into the full slab on every rank")
                out["config"]["graphs_per_gpu"] = hi - lo

    try:
        exchange_legs()
    except SystemExit:
        raise
    except Exception as ex:
        import traceback
        traceback.print_exc()
        out["exchange_legs_error"] = f"{type(ex).__name__}: {ex}"

    # throughput through the Dataset classes the trainers call (not the kernels): rank 0, N=1, ZINC-shaped workloads
    if rank == 0 and world == 1 and zinc and not args.no_boundary:
        try:                   # a secondary section must never cost the run its headline line
            out["boundary"] = boundary_section(d, G, dev, max_nodes, max_len)
        except Exception as ex:
            import traceback
            traceback.print_exc()
            out["boundary"] = {"error": f"{type(ex).__name__}: {ex}"}
    if torch_op is not None:
        out.setdefault("boundary", {})["torch_op_sent"] = torch_op

    # CPU baseline: the oracle (a port, not the reference's Python) on a bounded sample, rank 0, N=1 only
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not rows_u16:
        try:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as orc
            S = min(G, args.cpu_sample or (100000 if zinc else 4096))
            sl = lambda k, hi: None if k not in d else d[k][:hi]
            coo = orc.Coo(d["node_counts"][:S], d["edge_counts"][:S], d["src"][:int(host.edge_ptr[S])],
                          d["dst"][:int(host.edge_ptr[S])], sl("x", int(host.node_ptr[S])) if zinc else None,
                          sl("edge_attr", int(host.edge_ptr[S])) if zinc else None)
            cores = host_cores(orc.num_threads())
            okw = dict(labeled=zinc, num_node_types=ntypes, num_edge_types=etypes, remap_zinc=zinc, ld=ld, nthreads=cores)
            orc.sent(coo.slice(0, min(S, 2000)), max_nodes, max_len, 0, 0, **okw)
            reps, t0 = 0, time.perf_counter()
            while reps < 3 or time.perf_counter() - t0 < 10.0:
                ref, rln = orc.sent(coo, max_nodes, max_len, 0, args.warmup + reps, **okw)
                reps += 1
            cpu_s = (time.perf_counter() - t0) / reps
            # the sample doubles as an end-of-run parity check on the very buffers that were timed
            chk = min(S, 4096)
            step(args.warmup + reps - 1, scratch_len)
            torch.cuda.synchronize()
            same = np.array_equal(ids[:chk].cpu().numpy(), ref[:chk]) and np.array_equal(scratch_len[:chk].cpu().numpy(), rln[:chk])
            # the same restatement on ONE thread (SURVEY section 8d asks for both), on a smaller slice of the sample
            S1 = min(S, max(2000, int(S * 4 / max(cores, 4))))
            coo1 = coo.slice(0, S1)
            okw1 = dict(okw, nthreads=1)
            reps1, t0 = 0, time.perf_counter()
            while reps1 < 2 or time.perf_counter() - t0 < 4.0:
                _, rln1 = orc.sent(coo1, max_nodes, max_len, 0, reps1, **okw1)
                reps1 += 1
            cpu1_s = (time.perf_counter() - t0) / reps1
            out["cpu_baseline"] = dict(value=round(S / cpu_s, 1), unit="graphs/s", cores=cores, kind="port",
                                       sample=f"first {S} graphs of the same corpus, oracle/gtok_oracle.c:oracle_sent "
                                              f"(OpenMP, {cores} threads), {reps} passes",
                                       tokens_per_sec=round(float(rln.sum()) / cpu_s, 1), parity_with_gpu=bool(same),
                                       single_thread=dict(value=round(S1 / cpu1_s, 1), unit="graphs/s", cores=1,
                                                          tokens_per_sec=round(float(rln1.sum()) / cpu1_s, 1),
                                                          sample=f"first {S1} graphs of the same corpus, 1 thread, {reps1} passes"))
            if zinc and not args.no_ibtt:      # the IBTT serialiser's CPU restatement on the same sample
                lut_h = lut.cpu().numpy()
                orc.ibtt_zinc(coo.slice(0, min(S, 2000)), lut_h, max_len, vocab["<pad>"], ild, nthreads=cores)
                reps_i, t0 = 0, time.perf_counter()
                while reps_i < 3 or time.perf_counter() - t0 < 5.0:
                    iref, irln = orc.ibtt_zinc(coo, lut_h, max_len, vocab["<pad>"], ild, nthreads=cores)
                    reps_i += 1
                icpu = (time.perf_counter() - t0) / reps_i
                isame = np.array_equal(iids[:chk].cpu().numpy(), iref[:chk]) and np.array_equal(iln[:chk].cpu().numpy(), irln[:chk])
                out["ibtt"]["cpu_baseline"] = dict(value=round(S / icpu, 1), unit="graphs/s", cores=cores, kind="port",
                                                   sample=f"first {S} graphs of the same corpus, oracle/gtok_oracle.c:oracle_ibtt_zinc "
                                                          f"(OpenMP, {cores} threads), {reps_i} passes",
                                                   tokens_per_sec=round(float(irln.sum()) / icpu, 1), parity_with_gpu=bool(isame))
        except Exception as ex:     # the baseline is a reported figure: its failure must not cost the run its line
            import traceback
            traceback.print_exc()
            out.setdefault("cpu_baseline", {"error": f"{type(ex).__name__}: {ex}"})
    log("[bench] done")
    sys.stdout.flush()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
