"""Per-phase cycle counts of sent_blane_kernel on config-5 shaped batches.  Needs a library built with
-DGTOK_PHASE_TIMING and loaded through GTOK_LIB (profiling build: the last 8 columns of the row of every unit's lane 0
hold s_memtime sums for node selection / row load + counters + position token / bracket to visit-index space / bracket
tokens / row end + padding, the wave-level step and bracket-iteration counts, and the unit's life on the 100 MHz
clock).  Never ship that build.
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DGTOK_PHASE_TIMING -Iinclude \
        -o glearning-benchmark_amd/csrc/libgtok_prof.so glearning-benchmark_amd/csrc/gtok_sent.hip glearning-benchmark_amd/csrc/gtok_ibtt.hip
    GTOK_LIB=glearning-benchmark_amd/csrc/libgtok_prof.so python profiles/tools/phase_timing_blane.py"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
os.environ["GTOK_SENT_KERNEL"] = "blane"
for name, d in (("er", gtok.synth.er_batch_device(G, dev, seed=1000)), ("mix", gtok.synth.mix_batch_device(G, dev, seed=1000))):
    b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], device=dev)
    for k in range(3):
        ids, ln = gtok.ops.sent(b, b.max_nodes, 600, 0, k, ld=608)
    torch.cuda.synchronize()
    first = b.lane_order[::64].long() if b.lane_order is not None else torch.arange(0, G, 64, device=dev)
    ph = ids[first][:, -8:].cpu().numpy().astype(np.int64)
    names = ["select node", "row+counters+pos", "bracket -> visit space", "bracket tokens", "row end + pad"]
    tot = ph[:, :5].sum()
    print(f"== {name}: {ph.shape[0]} units")
    for i, nme in enumerate(names):
        print(f"{nme:24s} mean {ph[:, i].mean():10.0f} cycles  max {ph[:, i].max():10.0f}  share {ph[:, i].sum() / tot:.3f}")
    steps, it1, it2 = ph[:, 5], (ph[:, 6] >> 16) & 0xFFFF, ph[:, 6] & 0xFFFF
    print(f"steps mean {steps.mean():.1f} max {steps.max()}  bracket iterations mean {it1.mean():.1f} / {it2.mean():.1f} max {it1.max()} / {it2.max()}")
    print(f"cycles per step (select + row) {(ph[:, 0] + ph[:, 1]).sum() / steps.sum():.0f}   per bracket iteration {ph[:, 2].sum() / max(1, it1.sum()):.0f} / {ph[:, 3].sum() / max(1, it2.sum()):.0f}")
    life = (ph[:, 7] & 0xFFFFFFFF) / 100.0
    print("unit life us (min p10 p50 p90 max):", " ".join(f"{np.percentile(life, p):7.1f}" for p in (0, 10, 50, 90, 100)))
