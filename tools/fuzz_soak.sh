#!/bin/bash
# A long run of every mode of tools/fuzz_parity.py (the round's last build); one summary line per mode.
cd /root/repo
S=${1:-700}
run() { echo "## $*"; timeout 1500 "$@" 2>&1 | grep -E "^FAIL|^fuzz:|Error|error" | tail -12; }
run python tools/fuzz_parity.py --cases 2500 --seed $((S+1))
PTK_TEST_KNOBS=knn_cap_min_nq=1 run python tools/fuzz_parity.py --cases 1200 --seed $((S+2))
PTK_TEST_KNOBS=knn_cap_min_nq=1,knn_cap=2 run python tools/fuzz_parity.py --cases 600 --seed $((S+3))
run python tools/fuzz_parity.py --lninf --cases 800 --seed $((S+4))
run python tools/fuzz_parity.py --dtype float64 --lninf --cases 1000 --seed $((S+5))
PTK_TEST_KNOBS=knn_cap_min_nq=1,knn64_cap=1,radius64_cap=1 run python tools/fuzz_parity.py --dtype float64 --lninf --cases 1000 --seed $((S+10))
run python tools/fuzz_parity.py --topological --cases 800 --seed $((S+6))
run python tools/fuzz_parity.py --topological --dtype float64 --cases 800 --seed $((S+7))
run python tools/fuzz_parity.py --multi --cases 500 --seed $((S+8))
PTK_TEST_KNOBS=sort_block=1,p2_cap=2 run python tools/fuzz_parity.py --cases 800 --seed $((S+9))
