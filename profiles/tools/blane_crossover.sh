# where does the bit-matrix lane kernel overtake the wave-per-graph LDS kernel? (config-5 shaped batches)
for G in 4096 8192 16384 32768 65536; do python profiles/tools/time_sent_large.py $G 2>&1 | grep -v "order=0" | grep "G="; done
