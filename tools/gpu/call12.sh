#!/bin/bash
for A in 0 1 2 4 8 9 6 13 11 15; do echo "== HAB_OCP_ABLATE=$A (1 no reads, 2 no MFMA, 4 no epilogue, 8 no conversion)"; HAB_BF3=127 HAB_OCP_ABLATE=$A timeout 100 python tools/bench_layers.py 1024 2>&1 | grep "conv1 (obs ingest) fwd"; done
