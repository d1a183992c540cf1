#!/bin/bash
OUT=gpurun_out/${1:-r02d}; mkdir -p $OUT
echo "== prefetch default"; timeout 300 python3 tools/diag_exchange.py 2>&1 | tee $OUT/diag_default.log
echo "== prefetch off"; VGICP_PREFETCH_MARGIN=0 timeout 300 python3 tools/diag_exchange.py 2>&1 | tee $OUT/diag_nopf.log
