#!/bin/bash
# sweep GTOK_LANE_BLOCKS_PER_CU for the lane kernel on the ZINC-shaped corpus
for c in "$@"; do
  echo "blocks/CU cap $c"; GTOK_SENT_KERNEL=lane GTOK_LANE_BLOCKS_PER_CU=$c python3 profiles/tools/time_sent_zinc.py 2>/dev/null
done
