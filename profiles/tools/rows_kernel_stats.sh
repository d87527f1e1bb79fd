#!/bin/bash
# rocprofv3 kernel statistics of every kernel of gtok_rows.hip / gtok_csr.hip at ZINC-full scale: one bench.py run that carries the
# exchange legs (forced one-rank RCCL group: pack / unpack / gather), the boundary section (collate_batch, collate_epoch, text kernels)
# and the layout step, then the rows tests for the entry points no bench leg calls.
#   gpurun -- 'bash profiles/tools/rows_kernel_stats.sh'
export TMPDIR=/tmp
out=gpurun_out/rows_kernel_stats; rm -rf $out; mkdir -p $out
export GTOK_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -o s -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/bench.log 2>&1 || { tail -5 $out/bench.log; exit 1; }
unset GTOK_BENCH_FORCE_DIST RANK WORLD_SIZE LOCAL_RANK
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tests -o s -- python3 -m pytest tests/test_gpu_rows.py tests/test_gpu_boundary.py -q -x > $out/tests.log 2>&1 || { tail -5 $out/tests.log; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
w = csv.writer(sys.stdout)
for part, what in (("bench", "GTOK_BENCH_FORCE_DIST=1 ... rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline (ZINC-full: headline, row flavours, epoch loop, exchange legs, boundary section)"),
                   ("tests", "rocprofv3 --kernel-trace --stats -- python3 -m pytest tests/test_gpu_rows.py tests/test_gpu_boundary.py (test sizes: the entry points no bench leg calls)")):
    for f in glob.glob(f"{out}/{part}/**/*kernel_stats.csv", recursive=True):
        rows = list(csv.reader(open(f)))
        w.writerow([f"# {what}"]); w.writerow(rows[0])
        for r in rows[1:]:
            if "gtok" in r[0] or "csr_pack8" in r[0] or "adj_bits" in r[0]:
                w.writerow(r)
PY
find $out -name '*_kernel_trace.csv' -delete; find $out -name '*agent_info.csv' -delete
