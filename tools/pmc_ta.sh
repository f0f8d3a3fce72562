#!/bin/bash
# tools/pmc_ta.sh TAG CMD... -- the texture-addresser / vector-L1 counters of a command (every open
# question about the traversal kernels is about the gather path).  One counter set per rocprofv3 run.
# Output: gpurun_out/TAG_ta.txt (+ gpurun_out/TAG_ta_avail.txt: what `rocprofv3 -L` offers for TA / TCP / TD)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TCP|TD|TCC)_[A-Za-z0-9_]+" | sort -u > $R/gpurun_out/${TAG}_ta_avail.txt
i=0
# (the TA_* and TD_* sets stall rocprofv3 on this pool until the timeout kills the pass -- 4 minutes each, nothing
# collected: profiles/r05a_c3_ta_tcp_pmc.txt has no TA / TD rows for that reason; the TCP sets answer the questions)
for c in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
         "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
         "TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum" \
         "TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum"; do
  i=$((i+1))
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmcta_${TAG}_$i -o pmc -- "$@" > /dev/null 2>> $R/gpurun_out/pmcta_$TAG.log
  echo "ta pass $i ($c) rc=$?"
done
python $R/tools/rocprof_summary.py pmc /tmp/pmcta_${TAG}_*/pmc_results.db > $R/gpurun_out/${TAG}_ta.txt 2>&1
