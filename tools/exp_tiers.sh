#!/bin/bash
# Sweep of the narrow tiers of phase 2 (PTK_TIERS = cumulative per-mille marks of the ranked classes : lanes per wave).
for t in "60:4" "30:1" "60:1" "30:1,100:4" "30:1,150:4" "60:1,200:4" "30:1,100:4,300:16" "30:1,150:4,500:16" "60:1,200:4,600:16" "30:1,100:2,300:8,1000:32" "60:2,200:8,1000:32" "100:4,400:16,1000:32" "30:1,100:4,1000:16"; do
  echo "tiers=$t"
  PTK_TIERS=$t python tools/ab_knn1.py --variants 0 --rounds 3 2>&1 | grep -E "^0" | cut -c1-110
done
