#!/bin/bash
# Timing experiment: phase 2 with only the light / only the heavy wavefronts, for several heavy-class thresholds.
for hc in 4 5 6 7; do for mode in 0 1 2; do
  echo "heavy_class=$hc mode=$mode (0 all, 1 light only, 2 heavy only)"
  PTK_HEAVY_CLASS=$hc PTK_DEBUG_PHASE2=$mode python tools/ab_knn1.py --variants 0 --rounds 3 2>&1 | grep -E "^0" | cut -c1-80
done; done
