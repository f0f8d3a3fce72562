#!/usr/bin/env python3
"""Turns rocprofv3 (ROCm 7.2, rocpd SQLite output) results into the small text
summaries committed under profiles/.

    python tools/rocprof_summary.py stats gpurun_out/prof_x/knn1_results.db > profiles/r01_x_stats.txt
    python tools/rocprof_summary.py pmc   gpurun_out/pmc_x_*/pmc_results.db > profiles/r01_x_pmc.txt

`stats` reproduces what `rocprofv3 --kernel-trace --stats` reports (calls, total,
average, share per kernel) plus the per-dispatch resource columns of the trace.
`pmc` sums each collected counter per kernel and per dispatch.  FETCH_SIZE /
WRITE_SIZE are reported in the tool's native unit (KiB-like units of the derived
metric) AND converted to bytes; on gfx950 FETCH_SIZE under-counts wide (16 B per
lane) coalesced reads by 2x (MI355X_MICROARCH.md, HBM section) -- the corrected
value is printed next to the raw one and labelled.
"""

from __future__ import annotations

import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::", "rocprim::", name)
    m = re.search(r"(radix_sort_onesweep_\w+|exclusive_scan\w*|lookback_scan\w*|init_lookback\w*)", name)
    if "rocprim::" in name and m:
        return "rocprim::" + m.group(1)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def stats(path: str) -> None:
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(grid_x), max(workgroup_x) from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} "
          f"{'%':>6s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds_B':>7s} {'scr_B':>6s} {'grid':>9s} {'wg':>4s}")
    for r in rows:
        print(f"{short(r[0]):70s} {r[1]:6d} {r[2] / 1e6:10.3f} {r[3] / 1e3:10.2f} {r[4] / 1e3:10.2f} "
              f"{r[5] / 1e3:10.2f} {100 * r[2] / total:6.2f} {r[6]:5d} {r[7]:5d} {r[8]:5d} {r[9]:7d} {r[10]:6d} "
              f"{r[11]:9d} {r[12]:4d}")


def pmc(paths) -> None:
    print("# rocprofv3 --pmc summaries (one pass per file); values are sums over all dispatches of a kernel")
    for path in paths:
        db = sqlite3.connect(path)
        cur = db.cursor()
        cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
        name_col = "kernel_name" if "kernel_name" in cols else "name"
        rows = cur.execute(
            f"select {name_col}, counter_name, count(distinct dispatch_id), sum(value) "
            f"from counters_collection group by {name_col}, counter_name order by 4 desc").fetchall()
        print(f"\n## {path}")
        print(f"{'kernel':60s} {'counter':22s} {'dispatches':>10s} {'sum':>18s} {'per_dispatch':>18s}")
        for name, counter, nd, val in rows:
            if val is None or nd == 0:
                continue
            line = f"{short(name):60s} {counter:22s} {nd:10d} {val:18.1f} {val / nd:18.1f}"
            if counter in ("FETCH_SIZE", "WRITE_SIZE"):
                b = val / nd * 1024.0
                line += f"   = {b / 1e6:.1f} MB/dispatch"
                if counter == "FETCH_SIZE":
                    line += f" raw; x2 gfx950 wide-load correction = {2 * b / 1e6:.1f} MB"
            print(line)


def traffic(paths) -> None:
    """JSON: per kernel, HBM bytes per dispatch = 2 x FETCH_SIZE (gfx950 correction for
    16-byte-per-lane loads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, both x 1024
    (the counters are in KiB).  bench.py reads the committed file for `roofline.traffic`."""
    import json
    out = {}
    for path in paths:
        db = sqlite3.connect(path)
        cur = db.cursor()
        cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
        name_col = "kernel_name" if "kernel_name" in cols else "name"
        rows = cur.execute(
            f"select {name_col}, counter_name, count(distinct dispatch_id), sum(value) "
            f"from counters_collection where counter_name in ('FETCH_SIZE', 'WRITE_SIZE') "
            f"group by {name_col}, counter_name").fetchall()
        for name, counter, nd, val in rows:
            if not nd or val is None:
                continue
            k = out.setdefault(short(name), {})
            k[counter + "_raw_bytes_per_dispatch"] = val / nd * 1024.0
            k["dispatches_" + counter] = nd
    for k in out.values():
        f = k.get("FETCH_SIZE_raw_bytes_per_dispatch")
        w = k.get("WRITE_SIZE_raw_bytes_per_dispatch")
        if f is not None and w is not None:
            k["hbm_bytes_per_dispatch"] = 2.0 * f + w
    print(json.dumps({"note": "HBM bytes = 2 x FETCH_SIZE (gfx950 half-count of 16 B/lane loads) + WRITE_SIZE; "
                              "separate --pmc passes of the same bench.py command", "kernels": out}, indent=1))


def timeline(path: str) -> None:
    """Kernel sequence of ONE step of the headline search: between the second and third launch of phase 2 of the
    k = 1 search (bench.py runs the headline form first); without such launches, between the last two launches of
    the kernel with the largest total."""
    db = sqlite3.connect(path)
    rows = db.cursor().execute("select name, start, end from kernels order by start").fetchall()
    top = db.cursor().execute(
        "select name from kernels group by name order by sum(end-start) desc limit 1").fetchone()[0]
    p2 = [i for i, r in enumerate(rows) if "knn1_phase2_kernel" in r[0]]
    if len(p2) >= 3:
        i0, i1 = p2[1], p2[2]
    else:
        idx = [i for i, r in enumerate(rows) if r[0] == top]
        if len(idx) < 2:
            return
        i0, i1 = idx[-2], idx[-1]
    t0 = rows[i0][2]
    print(f"# one step of {path}: t = 0 at the end of the previous step's phase 2 (or dominant kernel)")
    print(f"{'start_us':>10s} {'dur_us':>10s}  kernel")
    for r in rows[i0 + 1:i1 + 1]:
        print(f"{(r[1] - t0) / 1e3:10.1f} {(r[2] - r[1]) / 1e3:10.1f}  {short(r[0])}")


if __name__ == "__main__":
    if len(sys.argv) < 3 or sys.argv[1] not in ("stats", "pmc", "timeline", "traffic"):
        raise SystemExit(__doc__)
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "timeline":
        timeline(sys.argv[2])
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[2:])
    else:
        pmc(sys.argv[2:])
