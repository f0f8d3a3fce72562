#!/usr/bin/env python3
"""When and where the wavefronts of the general traversal kernels run (what bounds knn = 16 and
the radius count pass -- throughput, or a tail of slow wavefronts?).

    python tools/wave_trace.py build            # here: tools/bin/libptk_trace.so = the library + -DPTK_WAVE_TRACE
    python tools/wave_trace.py run [L|U] [out]  # on the MI355X: BASELINE config 3, one traced launch of each kernel

The traced library is an experiment build; the shipped libptk.so has no trace code.  Per kernel the report gives the
launch span, the distribution of wavefront durations, how many wavefronts were resident over time (twenty slices of
the span), when each XCD went idle, and the share of the span in which fewer than half / a quarter of the wavefront
slots were taken.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "bin", "libptk_trace.so")


def build():
    from pico_tree_amd import build as b
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    # (all units in one command: an experiment build, compiled once)
    cmd = ["/opt/rocm/bin/hipcc"] + b.FLAGS + ["-shared", "-DPTK_WAVE_TRACE", "-I" + os.path.join(ROOT, "include"),
                                               "-I" + b.CSRC, "-o", LIB] + b.SOURCES
    subprocess.check_call(cmd)
    print(LIB)


def analyse(name, tr, slots):
    tr = tr[tr[:, 1] > 0]
    t0, t1 = tr[:, 0].astype(np.int64), tr[:, 1].astype(np.int64)
    start = t0.min()
    span = (t1.max() - start) / 100.0  # us (100 MHz)
    dur = (t1 - t0) / 100.0
    xcc = (tr[:, 2] >> 32).astype(np.int64) & 7
    cyc = tr[:, 3].astype(np.float64)
    rep = {"kernel": name, "wavefronts": int(len(tr)), "span_us": round(span, 1),
           "wave_us": {k: round(float(np.percentile(dur, p)), 1) for k, p in
                       (("p10", 10), ("p50", 50), ("p90", 90), ("p99", 99), ("p99.9", 99.9), ("max", 100))},
           "wave_us_mean": round(float(dur.mean()), 1),
           "mean_resident": round(float(dur.sum() / span), 1), "slots": slots,
           "shader_clock_mhz": round(float(np.median(cyc / np.maximum(dur, 1e-3))), 0)}
    edges = np.linspace(0, span, 21)
    res = []
    for a, b in zip(edges[:-1], edges[1:]):
        lo, hi = start + a * 100, start + b * 100
        ov = np.clip(np.minimum(t1, hi) - np.maximum(t0, lo), 0, None).sum() / max(hi - lo, 1)
        res.append(round(float(ov), 0))
    rep["resident_over_time"] = res
    # fine timeline: share of the span with fewer than half / a quarter of the slots taken
    ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t1, -np.ones_like(t1)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    level = np.cumsum(ev[:, 1])[:-1]
    dt = np.diff(ev[:, 0])
    tot = max(dt.sum(), 1)
    rep["share_below_half"] = round(float(dt[level < slots / 2].sum() / tot), 3)
    rep["share_below_quarter"] = round(float(dt[level < slots / 4].sum() / tot), 3)
    rep["xcd_idle_at_us"] = [round(float((t1[xcc == x].max() - start) / 100.0), 1) if (xcc == x).any() else None for x in range(8)]
    rep["xcd_waves"] = [int((xcc == x).sum()) for x in range(8)]
    # the last wavefronts to finish: how long they ran and when they started
    last = np.argsort(t1)[-5:]
    rep["last_to_finish"] = [{"start_us": round(float((t0[i] - start) / 100.0), 1), "ran_us": round(float(dur[i]), 1),
                              "block": int(i)} for i in last]
    # wavefronts in launch order: mean duration of each twentieth of the grid (is the expensive-first order working?)
    n = len(dur)
    rep["dur_by_block_twentieth_us"] = [round(float(dur[n * i // 20:n * (i + 1) // 20].mean()), 1) for i in range(20)]
    rep["start_by_block_twentieth_us"] = [round(float((t0[n * i // 20:n * (i + 1) // 20].mean() - start) / 100.0), 1) for i in range(20)]
    return rep


def run(cloud, out_path):
    import ctypes
    import torch
    import pico_tree_amd as pt
    from pico_tree_amd import datasets as ds

    pt._LIB_PATH = LIB
    lib = pt._load()
    lib.ptk_debug_wave_trace.restype = ctypes.c_int
    lib.ptk_debug_wave_trace.argtypes = [ctypes.c_void_p]
    pts, q = ds.config2_clouds(cloud)
    nq = len(q)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
    dq = torch.from_numpy(q).cuda()
    blocks = (nq + 63) // 64
    reports = []
    k = int(os.environ.get("TRACE_K", "16"))
    out = torch.empty((nq, k, 2), dtype=torch.int32, device="cuda")
    tree.search_knn(dq, k, out)
    torch.cuda.synchronize()
    trace = torch.zeros((blocks, 4), dtype=torch.int64, device="cuda")
    assert lib.ptk_debug_wave_trace(trace.data_ptr()) == 0
    tree.search_knn(dq, k, out)
    torch.cuda.synchronize()
    reports.append(analyse(f"knn_reg_kernel k={k} cloud {cloud}", trace.cpu().numpy().view(np.uint64), 256 * 20))
    print(json.dumps(reports[-1]), flush=True)
    # the cooperative search of the same step (selector 5; the general kernel writes the same words, so it runs untraced)
    lib.ptk_debug_wave_trace_select.argtypes = [ctypes.c_int]
    trace.zero_()
    assert lib.ptk_debug_wave_trace_select(5) == 0
    tree.search_knn(dq, k, out)
    torch.cuda.synchronize()
    tr = trace.cpu().numpy().view(np.uint64).copy()
    lib.ptk_debug_wave_trace_select(0)
    if (tr[:, 1] > 0).any():
        reports.append(analyse(f"knn_coop_kernel k={k} cloud {cloud}", tr, 8192))
        print(json.dumps(reports[-1]), flush=True)
    del out
    trace.zero_()
    res = tree.search_radius_device(dq, 1.0)
    torch.cuda.synchronize()
    trace.zero_()
    res = tree.search_radius_device(dq, 1.0)
    torch.cuda.synchronize()
    reports.append(analyse(f"radius_list_kernel r=1 cloud {cloud}", trace.cpu().numpy().view(np.uint64), 256 * 20))
    del res
    assert lib.ptk_debug_wave_trace(None) == 0
    with open(out_path, "w") as f:
        for r in reports:
            f.write(json.dumps(r) + "\n")
            print(json.dumps(r))


def run_k1(cloud, out_path, nq_cut=None):
    """The kernels of the two-phase k = 1 search, one traced step each (full batch, or its first nq_cut rows)."""
    import ctypes
    import torch
    import pico_tree_amd as pt
    from pico_tree_amd import datasets as ds

    pt._LIB_PATH = LIB
    lib = pt._load()
    lib.ptk_debug_wave_trace.argtypes = [ctypes.c_void_p]
    lib.ptk_debug_wave_trace_select.argtypes = [ctypes.c_int]
    pts, q = ds.config2_clouds(cloud)
    if nq_cut:
        q = np.ascontiguousarray(q[:nq_cut])
    nq = len(q)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
    dq = torch.from_numpy(q).cuda()
    out = torch.empty((nq, 1, 2), dtype=torch.int32, device="cuda")
    for _ in range(3):
        tree.search_knn(dq, 1, out)
    torch.cuda.synchronize()
    blocks = nq // 64 + 8192
    trace = torch.zeros((blocks, 4), dtype=torch.int64, device="cuda")
    assert lib.ptk_debug_wave_trace(trace.data_ptr()) == 0
    names = {1: "knn1_phase1u_kernel", 2: "knn1_phase2_kernel", 3: "knn1_coop_kernel<direct>", 4: "knn1_coop_kernel<tail>"}
    slots = {1: 256 * 32, 2: 256 * 26, 3: 256 * 17, 4: 256 * 17}
    with open(out_path, "w") as f:
        for sel in (1, 2, 3, 4):
            trace.zero_()
            assert lib.ptk_debug_wave_trace_select(sel) == 0
            tree.search_knn(dq, 1, out)
            torch.cuda.synchronize()
            tr = trace.cpu().numpy().view(np.uint64)
            if not (tr[:, 1] > 0).any():
                continue
            r = analyse(f"{names[sel]} k=1 cloud {cloud} nq={nq}", tr, slots[sel])
            f.write(json.dumps(r) + "\n")
            print(json.dumps(r), flush=True)
    lib.ptk_debug_wave_trace_select(0)
    lib.ptk_debug_wave_trace(None)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "k1":
        run_k1(sys.argv[2] if len(sys.argv) > 2 else "L",
               sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "r05_wave_trace_k1.jsonl"),
               int(sys.argv[4]) if len(sys.argv) > 4 else None)
    else:
        run(sys.argv[2] if len(sys.argv) > 2 else "L",
            sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "r05_wave_trace.jsonl"))
