#!/bin/bash
# usage: sweep_env.sh VAR v1 v2 ... -- extra bench args ; prints kernel_ms per value
var=$1; shift
vals=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do vals+=("$1"); shift; done
shift
for v in "${vals[@]}"; do
  env $var=$v python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ibtt "$@" 2>/dev/null > /tmp/sweep.json
  python3 - "$var" "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/sweep.json"))
print(sys.argv[1], sys.argv[2], "kernel_ms", d["roofline"]["kernel_ms"], "graphs/s", d["value"])
PY
done
