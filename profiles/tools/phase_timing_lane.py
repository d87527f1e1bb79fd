"""Per-phase cycle counts of sent_lane_kernel on the ZINC-shaped corpus.  Needs a library built with
-DGTOK_PHASE_TIMING and loaded through GTOK_LIB (profiling build: the last 8 columns of every unit's first row hold
s_memtime deltas for staging / counter init / walk / row end, the wave's start and end on the 100 MHz real-time
clock, the walk's iteration count and the workgroup id).  Never ship that build.
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DGTOK_PHASE_TIMING -Iinclude \
        -o glearning-benchmark_amd/csrc/libgtok_prof.so glearning-benchmark_amd/csrc/gtok_sent.hip glearning-benchmark_amd/csrc/gtok_ibtt.hip
    GTOK_LIB=glearning-benchmark_amd/csrc/libgtok_prof.so python profiles/tools/phase_timing_lane.py"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 249456
d = gtok.synth.zinc_like(G, seed=1000)
host = gtok.GraphBatch.from_coo(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"])
b = host.to(dev)
ld = 216
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
pk = gtok.ops.PackedRows(G, G * 104, False, dev) if os.environ.get("GTOK_PT_PACKED") == "1" else None     # gtok_sent_packed: the copy's cycles too
for k in range(3):
    ids, ln = gtok.ops.sent(b, 37, 1024, 0, k, ld=ld, packed=pk, **kw)
torch.cuda.synchronize()
assert int(ln.max()) <= ld - 8
sb = b.lane_sorted          # the reordered copy ops.sent walks by default (GTOK_NO_LANE_SORT=1: the batch as stored)
first = torch.arange(0, G, 64, device=dev) if sb is None else sb.graph_ids[sb.unit_ptr[:-1].long()].long()
print("units", first.numel(), "reordered" if sb is not None else "dataset order")
ph = ids[first][:, -8:].cpu().numpy().astype(np.int64)
names = ["staging", "counter init", "walk", "row end"]
tot = ph[:, :4].sum()
for i, nme in enumerate(names):
    print(f"{nme:14s} mean {ph[:, i].mean():10.0f} cycles  max {ph[:, i].max():10.0f}  share {ph[:, i].sum() / tot:.3f}")
print("mean cycles per unit", ph[:, :4].sum(1).mean(), " iterations mean", ph[:, 6].mean(), "max", ph[:, 6].max(),
      " walk cycles per iteration", ph[:, 2].sum() / ph[:, 6].sum())
if pk is not None:
    assert pk.fused and int(pk.status()) == 0
    print(f"gtok_sent_packed (inside 'row end'): until the atomic has answered mean {ph[:, 6].mean():.0f} cycles (max {ph[:, 6].max()}), copy mean {ph[:, 7].mean():.0f} (max {ph[:, 7].max()})")
rt0 = (ph[:, 4] & 0xFFFFFFFF); rt1 = (ph[:, 5] & 0xFFFFFFFF)
base = rt0.min()
s0, s1 = (rt0 - base) / 100.0, (rt1 - base) / 100.0          # us on the 100 MHz clock
q = lambda a: " ".join(f"{np.percentile(a, p):7.1f}" for p in (0, 10, 50, 90, 100))
print("wave start us (min p10 p50 p90 max):", q(s0))
print("wave end   us (min p10 p50 p90 max):", q(s1))
print("wave life  us (min p10 p50 p90 max):", q(s1 - s0))
