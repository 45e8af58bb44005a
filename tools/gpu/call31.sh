#!/bin/bash
for V in 0 1; do echo "== HAB_BF3_TALL64=$V"; HAB_BF3_TALL64=$V timeout 100 python tools/bench_layers.py 512 > /tmp/o.txt 2>&1; grep -E "conv2|l2 " /tmp/o.txt | grep -E "fwd|dgrad"; tail -3 /tmp/o.txt | cut -c1-300; done
for V in 0 1 0 1; do HAB_BF3_TALL64=$V timeout 200 python bench.py --no-cpu-baseline 2>/tmp/e.txt | grep -o '"value": [0-9.]*' | head -1; tail -2 /tmp/e.txt | cut -c1-300; done
