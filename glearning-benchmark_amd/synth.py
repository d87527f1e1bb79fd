"""Synthetic corpora for parity tests and bench.py (no dataset exists offline: SURVEY.md F4).

  zinc_like()        molecules shaped like PyG ZINC: random tree + ring
                     closures, 28-way atom ids skewed to C, bond ids 1..3 with rare 0/4, so that the
                     reference's fallback branches ('X', 'unknown', 22+token_id) fire.
  graph_token_like() graphs in the families docs/synthetic_data.md:10-20 lists, with the task text
                     docs/synthetic_data.md:46-68 describes.
Both return plain numpy batched COO (dict), edge lists in the order a PyG / graph-token file would
list them.
"""
from typing import Dict, List, Optional, Sequence

import numpy as np


def _ptr(counts: np.ndarray) -> np.ndarray:
    p = np.zeros(counts.size + 1, np.int64)
    np.cumsum(counts, out=p[1:])
    return p


def zinc_like(num_graphs: int, seed: int = 0, mean_nodes: float = 23.2, std_nodes: float = 4.5,
              min_nodes: int = 9, max_nodes: int = 37, coalesced: bool = True) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    G = int(num_graphs)
    n = np.clip(np.rint(rng.normal(mean_nodes, std_nodes, G)), min_nodes, max_nodes).astype(np.int64)
    nptr = _ptr(n)
    N = int(nptr[-1])
    gid = np.repeat(np.arange(G), n)
    li = np.arange(N) - nptr[gid]                                  # local node index
    # spanning tree: mostly chains with occasional branches
    r = rng.random(N)
    parent = np.where(r < 0.72, li - 1, np.floor(rng.random(N) * np.maximum(li, 1)).astype(np.int64))
    has_p = li >= 1
    tu, tv, tg = parent[has_p], li[has_p], gid[has_p]
    # ring closures: 1..4 extra bonds a -- a+d (d = 2..6) per molecule
    k = rng.integers(1, 5, G)
    rg = np.repeat(np.arange(G), k)
    a = np.floor(rng.random(rg.size) * n[rg]).astype(np.int64)
    b = a + rng.integers(2, 7, rg.size)
    ok = b < n[rg]
    ru, rv, rgg = a[ok], b[ok], rg[ok]
    u = np.concatenate([tu, ru]); v = np.concatenate([tv, rv]); g = np.concatenate([tg, rgg])
    lo, hi = np.minimum(u, v), np.maximum(u, v)
    key = (g * 64 + lo) * 64 + hi
    key = np.unique(key)                                          # simple graph
    g, lo, hi = key // 4096, (key // 64) % 64, key % 64
    bond = rng.choice(np.array([1, 2, 3, 0, 4]), size=key.size, p=[0.74, 0.19, 0.05, 0.01, 0.01])
    # both directions, like PyG's undirected edge_index
    src = np.concatenate([lo, hi]); dst = np.concatenate([hi, lo]); gg = np.concatenate([g, g])
    ea = np.concatenate([bond, bond])
    if coalesced:
        order = np.lexsort((dst, src, gg))                        # row-sorted within each graph
    else:
        order = np.lexsort((rng.random(gg.size), gg))             # arbitrary order inside a graph
    src, dst, gg, ea = src[order], dst[order], gg[order], ea[order]
    ecount = np.bincount(gg, minlength=G).astype(np.int64)
    p_atom = np.full(28, 0.02 / 19)
    p_atom[:9] = [0.70, 0.10, 0.10, 0.02, 0.005, 0.03, 0.015, 0.005, 0.005]
    p_atom /= p_atom.sum()
    x = rng.choice(28, size=N, p=p_atom)
    y = rng.normal(0.0, 2.0, G).astype(np.float32)
    return dict(node_counts=n, edge_counts=ecount, src=src.astype(np.int64), dst=dst.astype(np.int64),
                x=x.astype(np.int64), edge_attr=ea.astype(np.int64), y=y)


# ------------------------------------------------------------------------------------------------
def _family_edges(alg: str, n: int, rng: np.random.Generator, p: float) -> np.ndarray:
    """Undirected simple edge list [m,2] with u<v, in adjacency-iteration order (networkx edges())."""
    iu, iv = np.triu_indices(n, 1)
    if alg == "er":
        keep = rng.random(iu.size) < p
    elif alg == "complete":
        keep = np.ones(iu.size, bool)
    elif alg == "path":
        keep = iv == iu + 1
    elif alg == "star":
        keep = iu == 0
    elif alg == "sbm":
        blk = (np.arange(n) * 2) // max(n, 1)
        same = blk[iu] == blk[iv]
        keep = rng.random(iu.size) < np.where(same, min(1.0, 2.0 * p), 0.25 * p)
    elif alg in ("ba", "sfn"):
        m = max(1, int(round(p * (n - 1) / 2)))
        deg = np.zeros(n); edges = []
        for t in range(1, n):
            w = deg[:t] + 1.0
            tgt = rng.choice(t, size=min(m, t), replace=False, p=w / w.sum())
            for s in tgt:
                edges.append((int(s), t)); deg[s] += 1; deg[t] += 1
        e = np.array(sorted(set(edges)), np.int64).reshape(-1, 2)
        return e
    else:
        raise ValueError(f"unknown algorithm {alg}")
    return np.stack([iu[keep], iv[keep]], 1).astype(np.int64)


def _has_cycle(n: int, e: np.ndarray) -> bool:
    par = list(range(n))

    def find(a):
        while par[a] != a:
            par[a] = par[par[a]]
            a = par[a]
        return a
    for u, v in e:
        ru, rv = find(int(u)), find(int(v))
        if ru == rv:
            return True
        par[ru] = rv
    return False


def _bfs_dist(n: int, e: np.ndarray, s: int) -> np.ndarray:
    adj: List[List[int]] = [[] for _ in range(n)]
    for u, v in e:
        adj[int(u)].append(int(v)); adj[int(v)].append(int(u))
    d = np.full(n, -1); d[s] = 0; q = [s]
    for a in q:
        for b in adj[a]:
            if d[b] < 0:
                d[b] = d[a] + 1; q.append(b)
    return d


def graph_token_like(num_graphs: int, seed: int = 1234, algorithms: Sequence[str] = ("er", "ba", "sbm", "path", "star", "complete"),
                     min_nodes: int = 10, max_nodes: int = 49, min_sparsity: float = 0.1, max_sparsity: float = 0.2,
                     task: Optional[str] = "cycle_check", with_text: bool = True) -> Dict[str, object]:
    """Per-graph Python loop: fine up to ~1e5 graphs.  task in {None,'cycle_check','shortest_path'}."""
    rng = np.random.default_rng(seed)
    ncs, ecs, srcs, dsts, texts, labels, queries, algs = [], [], [], [], [], [], [], []
    for i in range(num_graphs):
        alg = algorithms[i % len(algorithms)]
        n = int(rng.integers(min_nodes, max_nodes + 1))
        e = _family_edges(alg, n, rng, float(rng.uniform(min_sparsity, max_sparsity)))
        ncs.append(n); ecs.append(e.shape[0]); srcs.append(e[:, 0]); dsts.append(e[:, 1]); algs.append(alg)
        q, lab, tail = None, None, ""
        if task == "cycle_check":
            lab = int(_has_cycle(n, e)); tail = f"<q> has_cycle <p> {'yes' if lab else 'no'} <eos>"
        elif task == "shortest_path":
            u = int(rng.integers(0, n)); v = int(rng.integers(0, n - 1)); v += v >= u
            d = int(_bfs_dist(n, e, u)[v])
            q = (u, v); lab = d - 1 if d > 0 else None
            tail = f"<q> shortest_distance {u} {v} <p> {'len%d' % d if d > 0 else 'INF'} <eos>"
        labels.append(lab); queries.append(q)
        if with_text:
            body = " ".join(f"{a} {b} <e>" for a, b in e)
            texts.append(" ".join(t for t in ("<bos>", body, "<n>", " ".join(map(str, range(n))), tail) if t))
    cat = lambda l: np.concatenate(l) if l else np.zeros(0, np.int64)
    return dict(node_counts=np.asarray(ncs, np.int64), edge_counts=np.asarray(ecs, np.int64), src=cat(srcs),
                dst=cat(dsts), texts=texts, labels=labels, queries=queries, algorithms=algs)


def er_batch(num_graphs: int, seed: int = 0, min_nodes: int = 10, max_nodes: int = 256,
             min_sparsity: float = 0.1, max_sparsity: float = 0.2, chunk: int = 2048) -> Dict[str, np.ndarray]:
    """Vectorised Erdős–Rényi corpus for the large-graph roofline run:
    one direction per undirected edge (u<v), row-sorted, as graph-token files list them."""
    rng = np.random.default_rng(seed)
    n = rng.integers(min_nodes, max_nodes + 1, num_graphs).astype(np.int64)
    p = rng.uniform(min_sparsity, max_sparsity, num_graphs)
    srcs, dsts, ecs = [], [], np.zeros(num_graphs, np.int64)
    for c0 in range(0, num_graphs, chunk):
        nn, pp = n[c0:c0 + chunk], p[c0:c0 + chunk]
        m = int(nn.max())
        iu, iv = np.triu_indices(m, 1)
        for j in range(nn.size):
            sel = iv < nn[j]
            keep = rng.random(int(sel.sum())) < pp[j]
            srcs.append(iu[sel][keep]); dsts.append(iv[sel][keep]); ecs[c0 + j] = int(keep.sum())
    cat = lambda l: np.concatenate(l).astype(np.int64) if l else np.zeros(0, np.int64)
    return dict(node_counts=n, edge_counts=ecs, src=cat(srcs), dst=cat(dsts))


def er_batch_device(num_graphs: int, device, seed: int = 0, min_nodes: int = 10, max_nodes: int = 256,
                    min_sparsity: float = 0.1, max_sparsity: float = 0.2, chunk: int = 512) -> Dict[str, np.ndarray]:
    """Same corpus family as er_batch(), sampled on the GPU with torch (bench.py's large-graph workload:
    the host loop would take minutes at 10^5 graphs).  Returns host numpy batched COO, u<v, row-sorted."""
    import torch
    gen = torch.Generator(device=device); gen.manual_seed(seed)
    n = torch.randint(min_nodes, max_nodes + 1, (num_graphs,), generator=gen, device=device)
    p = torch.rand((num_graphs,), generator=gen, device=device) * (max_sparsity - min_sparsity) + min_sparsity
    M = max_nodes
    iu = torch.arange(M, device=device)
    upper = iu[None, :] > iu[:, None]
    srcs, dsts, counts = [], [], []
    for c0 in range(0, num_graphs, chunk):
        nn, pp = n[c0:c0 + chunk], p[c0:c0 + chunk]
        keep = torch.rand((nn.numel(), M, M), generator=gen, device=device) < pp[:, None, None]
        keep &= upper[None] & (iu[None, None, :] < nn[:, None, None])
        g, u, v = keep.nonzero(as_tuple=True)          # sorted by (g, u, v)
        counts.append(torch.bincount(g, minlength=nn.numel()))
        srcs.append(u.to(torch.int32)); dsts.append(v.to(torch.int32))
    cat = lambda l: torch.cat(l).cpu().numpy()
    return dict(node_counts=n.cpu().numpy().astype(np.int64), edge_counts=cat(counts).astype(np.int64),
                src=cat(srcs).astype(np.int64), dst=cat(dsts).astype(np.int64))


def graph_token_tree(num_graphs: int, seed: int = 1234, task: str = "cycle_check",
                     algorithms: Sequence[str] = ("er", "ba", "sbm", "path", "star", "complete"),
                     splits: Sequence[str] = ("train", "test"), min_nodes: int = 10, max_nodes: int = 49,
                     pairs_per_graph: int = 1, use_split_tasks_dirs: bool = True) -> Dict[str, list]:
    """A graph-token task directory as {relative path: list of records} (what graph_task_generator.py leaves on
    disk, README.md:29-35 / docs/synthetic_data.md:73-128): `tasks_{train,test}/<task>/<algo>/<split>/<i>.json`,
    one file per graph, one record per query — shortest_path files carry `pairs_per_graph` records over the same
    graph (unreachable pairs are written with `INF`, which the loaders skip).  `num_graphs` graphs per
    (algorithm, split)."""
    rng = np.random.default_rng(seed)
    tree: Dict[str, list] = {}
    for split in splits:
        top = ("tasks_test" if split in ("val", "test") else "tasks_train") if use_split_tasks_dirs else "tasks"
        for alg in algorithms:
            for i in range(num_graphs):
                n = int(rng.integers(min_nodes, max_nodes + 1))
                e = _family_edges(alg, n, rng, float(rng.uniform(0.1, 0.2)))
                body = " ".join(f"{a} {b} <e>" for a, b in e)
                head = " ".join(t for t in ("<bos>", body, "<n>", " ".join(map(str, range(n)))) if t)
                recs = []
                if task == "cycle_check":
                    recs.append({"text": f"{head} <q> has_cycle <p> {'yes' if _has_cycle(n, e) else 'no'} <eos>"})
                else:
                    for _ in range(pairs_per_graph):
                        u = int(rng.integers(0, n)); v = int(rng.integers(0, n - 1)); v += v >= u
                        d = int(_bfs_dist(n, e, u)[v])
                        recs.append({"text": f"{head} <q> shortest_distance {u} {v} <p> {'len%d' % d if d > 0 else 'INF'} <eos>"})
                tree[f"{top}/{task}/{alg}/{split}/{i:05d}.json"] = recs
    return tree


def write_tree(root: str, tree: Dict[str, list]) -> None:
    import json
    import os
    for rel, recs in tree.items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(recs, f)


MIX_FAMILIES = ("er", "ba", "sbm", "sfn", "path", "star", "complete")


def mix_batch_device(num_graphs: int, device, seed: int = 0, min_nodes: int = 10, max_nodes: int = 256,
                     min_sparsity: float = 0.1, max_sparsity: float = 0.2, families: Sequence[str] = MIX_FAMILIES,
                     chunk: int = 512) -> Dict[str, np.ndarray]:
    """BASELINE config 5's corpus shape (graph_generator.sh families er / ba / sbm / sfn / complete / star / path,
    docs/synthetic_data.md:86, 10..max_nodes nodes, sparsity 0.1-0.2), sampled on the GPU with torch: graph i is of
    family families[i % len(families)], so every contiguous shard holds the whole mix.  One direction per
    undirected edge (u < v), row-sorted, as graph-token files list them.  Returns host numpy batched COO plus
    `family` (index into `families`) per graph.  ba / sfn: preferential attachment, m = round(p (n-1) / 2) >= 1 new
    edges per arriving node, drawn without replacement with probability ~ degree + 1."""
    import torch
    gen = torch.Generator(device=device); gen.manual_seed(seed)
    F = len(families)
    n = torch.randint(min_nodes, max_nodes + 1, (num_graphs,), generator=gen, device=device)
    p = torch.rand((num_graphs,), generator=gen, device=device) * (max_sparsity - min_sparsity) + min_sparsity
    fam = torch.arange(num_graphs, device=device) % F
    M = max_nodes
    iu = torch.arange(M, device=device)
    upper = iu[None, :] > iu[:, None]                                 # [u, v]: u < v
    fid = {name: k for k, name in enumerate(families)}
    srcs, dsts, counts = [], [], []
    for c0 in range(0, num_graphs, chunk):
        nn, pp, ff = n[c0:c0 + chunk], p[c0:c0 + chunk], fam[c0:c0 + chunk]
        C = nn.numel()
        r = torch.rand((C, M, M), generator=gen, device=device)
        keep = torch.zeros((C, M, M), dtype=torch.bool, device=device)
        is_ = lambda name: (ff == fid[name])[:, None, None] if name in fid else torch.zeros((C, 1, 1), dtype=torch.bool, device=device)
        keep |= is_("er") & (r < pp[:, None, None])
        keep |= is_("complete")
        keep |= is_("path") & (iu[None, None, :] == iu[None, :, None] + 1)
        keep |= is_("star") & (iu[None, :, None] == 0)
        blk = (iu[None, :] * 2) // nn[:, None].clamp(min=1)           # two communities
        same = blk[:, :, None] == blk[:, None, :]
        keep |= is_("sbm") & (r < torch.where(same, (2.0 * pp).clamp(max=1.0)[:, None, None], (0.25 * pp)[:, None, None]))
        pa = torch.zeros(C, dtype=torch.bool, device=device)
        for name in ("ba", "sfn"):
            if name in fid:
                pa |= ff == fid[name]
        if bool(pa.any()):
            idx = pa.nonzero(as_tuple=True)[0]
            na, ma = nn[idx], torch.clamp(torch.round(pp[idx] * (nn[idx] - 1).float() / 2).long(), min=1)
            A = idx.numel()
            mmax = int(ma.max())
            deg = torch.zeros((A, M), device=device)
            sub = torch.zeros((A, M, M), dtype=torch.bool, device=device)
            ar = torch.arange(A, device=device)
            for t in range(1, int(na.max())):
                w = deg[:, :t] + 1.0
                k = min(mmax, t)
                tgt = torch.multinomial(w, k, replacement=False, generator=gen)          # [A, k] nodes < t
                use = (torch.arange(k, device=device)[None, :] < torch.minimum(ma, torch.tensor(t, device=device))[:, None]) \
                    & (t < na)[:, None]
                a_i = ar[:, None].expand(A, k)[use]; s_i = tgt[use]
                sub[a_i, s_i, t] = True
                deg[a_i, s_i] += 1.0
                deg[:, t] += use.sum(1).float()
            keep[idx] = sub
        keep &= upper[None] & (iu[None, None, :] < nn[:, None, None])
        g, u, v = keep.nonzero(as_tuple=True)                         # sorted by (g, u, v)
        counts.append(torch.bincount(g, minlength=C))
        srcs.append(u.to(torch.int32)); dsts.append(v.to(torch.int32))
    cat = lambda l: torch.cat(l).cpu().numpy()
    return dict(node_counts=n.cpu().numpy().astype(np.int64), edge_counts=cat(counts).astype(np.int64),
                src=cat(srcs).astype(np.int64), dst=cat(dsts).astype(np.int64), family=fam.cpu().numpy())


# ------------------------------------------------------------------------------------------------
class _Bag:
    """Attribute bag standing in for torch_geometric.data.Data."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class InMemoryLike:
    """A split held the way torch_geometric's InMemoryDataset holds one (what torch_geometric.datasets.ZINC is,
    zinc_dataset_autograph.py:44 / zinc_dataset_indexbase.py:79; no torch_geometric offline):
    `_data` = every attribute concatenated over the graphs (x [sum N, 1], edge_index [2, sum E] with LOCAL node ids,
    edge_attr [sum E], y [G]), `slices[name]` = the per-graph offsets, `_indices` = an optional subset, items
    separated on demand (a fresh object per fetch, like InMemoryDataset.get's copy).  Built from a batched-COO dict
    (zinc_like())."""

    def __init__(self, d: Dict[str, np.ndarray], indices: Optional[Sequence[int]] = None, transform=None):
        import torch
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
        self._data = _Bag(x=t(d["x"], torch.long).view(-1, 1), edge_index=torch.stack([t(d["src"], torch.long), t(d["dst"], torch.long)]),
                          edge_attr=t(d["edge_attr"], torch.long), y=t(d["y"], torch.float32))
        nptr, eptr = t(_ptr(np.asarray(d["node_counts"])), torch.long), t(_ptr(np.asarray(d["edge_counts"])), torch.long)
        self.slices = {"x": nptr, "edge_index": eptr, "edge_attr": eptr, "y": torch.arange(len(d["node_counts"]) + 1)}
        self._indices = None if indices is None else list(indices)
        self.transform = transform
        self._n, self._e = nptr.tolist(), eptr.tolist()

    def indices(self):
        return range(len(self._n) - 1) if self._indices is None else self._indices

    def __len__(self):
        return len(self.indices())

    def __getitem__(self, idx):
        g = self.indices()[idx]
        n0, n1, e0, e1 = self._n[g], self._n[g + 1], self._e[g], self._e[g + 1]
        dt = self._data
        item = _Bag(x=dt.x[n0:n1], edge_index=dt.edge_index[:, e0:e1], edge_attr=dt.edge_attr[e0:e1], y=dt.y[g:g + 1],
                    num_nodes=n1 - n0)
        return item if self.transform is None else self.transform(item)
