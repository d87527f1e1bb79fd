# workgroup shape / priority variants of sent_blane_kernel on the config-5 shaped batches
for v in "8 1" "8 0" "4 0"; do set -- $v; echo "waves per workgroup $1, priorities $2"; GTOK_BLANE_WAVES=$1 GTOK_BLANE_PRIO=$2 python profiles/tools/time_sent_large.py ${G:-125000} 2>&1 | grep "blane.*order=1"; done
