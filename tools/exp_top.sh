#!/bin/bash
for deal in 0 1; do for hc in 4 5; do for pm in 30 100 300; do for hl in 4 8; do
  echo "deal=$deal heavy_class=$hc top_permille=$pm top_lanes=$hl"
  PTK_DEAL=$deal PTK_HEAVY_CLASS=$hc PTK_TOP_PERMILLE=$pm PTK_TOP_LANES=$hl python tools/ab_knn1.py --variants 0 --rounds 3 2>&1 | grep -E "^0" | cut -c1-110
done; done; done; done
