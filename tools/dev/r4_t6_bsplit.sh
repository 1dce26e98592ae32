#!/bin/bash
# t6: what would pre-split B planes (delivered by LDS-DMA) save at most?  PGASR_TN_DIAG bit 2 (4): B neither split nor stored; bit 3 (8): nor loaded.
for d in 0 4 12 1 0; do echo "== PGASR_TN_DIAG=$d"; PGASR_TN_DIAG=$d QUICK=1 python tools/dev/tools_gemm6.py 2>&1 | grep "dW ="; done
