"""GraphTokenDatasetForAutoGraph.process() over a synthetic graph-token tree: host parsers vs the device text parser."""
import importlib, os, sys, tempfile, time, contextlib, io
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
gdl = gtok.graph_data_loader
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
with tempfile.TemporaryDirectory() as root:
    t0 = time.perf_counter()
    tree = gtok.synth.graph_token_tree(n, seed=3, task="cycle_check", algorithms=("er", "ba", "sbm"), splits=("train",), min_nodes=10, max_nodes=49)
    gtok.synth.write_tree(root, tree)
    print(f"{len(tree)} files written in {time.perf_counter() - t0:.1f} s")
    kw = dict(root=root, task="cycle_check", algorithm=["er", "ba", "sbm"], split="train", use_cache=False)
    G = gdl.GraphTokenDatasetForAutoGraph
    for name, thr in (("device parser", 256), ("host parsers", 10 ** 9), ("device parser", 256)):
        G.DEVICE_PARSE_MIN = thr
        with contextlib.redirect_stdout(io.StringIO()):
            t0 = time.perf_counter(); ds = G(**kw); dt = time.perf_counter() - t0
        print(f"{name:14s}: {len(ds)} items in {dt:.2f} s = {len(ds) / dt:.0f} items/s")
