#!/usr/bin/env python3
"""What the host link and the host cores of this box can do (sizes of BASELINE config 2, k = 1):
pageable vs pinned copies both ways, host memcpy into a pinned block with 1 .. N threads.  One JSON line."""
from __future__ import annotations

import json
import os
import subprocess
import threading
import time

import numpy as np
import torch


def rate(nbytes, fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return round(nbytes * reps / (time.perf_counter() - t0) / 1e9, 2)


def main():
    up, down = 86_410_356, 57_606_904
    res = {}
    h_up = torch.empty(up, dtype=torch.uint8)
    h_up.fill_(1)
    p_up = torch.empty(up, dtype=torch.uint8, pin_memory=True)
    p_up.fill_(1)
    d_up = torch.empty(up, dtype=torch.uint8, device="cuda")
    h_dn = torch.empty(down, dtype=torch.uint8)
    p_dn = torch.empty(down, dtype=torch.uint8, pin_memory=True)
    d_dn = torch.empty(down, dtype=torch.uint8, device="cuda")
    res["h2d_pageable_GBs"] = rate(up, lambda: d_up.copy_(h_up))
    res["h2d_pinned_GBs"] = rate(up, lambda: d_up.copy_(p_up, non_blocking=True))
    res["d2h_pageable_GBs"] = rate(down, lambda: h_dn.copy_(d_dn))
    res["d2h_pinned_GBs"] = rate(down, lambda: p_dn.copy_(d_dn, non_blocking=True))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def both():
        with torch.cuda.stream(s1):
            d_up.copy_(p_up, non_blocking=True)
        with torch.cuda.stream(s2):
            p_dn.copy_(d_dn, non_blocking=True)
    res["duplex_pinned_GBs_total"] = rate(up + down, both)
    # pinned copies in pieces (what a pipeline would issue)
    for piece_mb in (2, 8, 32):
        piece = piece_mb << 20

        def pieces():
            for lo in range(0, up, piece):
                d_up[lo:lo + piece].copy_(p_up[lo:lo + piece], non_blocking=True)
        res[f"h2d_pinned_{piece_mb}MB_pieces_GBs"] = rate(up, pieces)
    # host memcpy pageable -> pinned with T threads
    src = h_up.numpy()
    dst = p_up.numpy()
    for threads in (1, 2, 4, 8, 16, 32):
        bounds = np.linspace(0, up, threads + 1).astype(np.int64)

        def run():
            ts = [threading.Thread(target=lambda a, b: np.copyto(dst[a:b], src[a:b]), args=(bounds[i], bounds[i + 1]))
                  for i in range(threads)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        run()
        t0 = time.perf_counter()
        for _ in range(5):
            run()
        res[f"memcpy_{threads}thr_GBs"] = round(up * 5 / (time.perf_counter() - t0) / 1e9, 2)
    try:
        res["lscpu"] = {ln.split(":")[0].strip(): ln.split(":", 1)[1].strip()
                        for ln in subprocess.check_output(["lscpu"], text=True).splitlines()
                        if ln.split(":")[0].strip() in ("Model name", "Socket(s)", "Core(s) per socket", "Thread(s) per core",
                                                        "CPU(s)", "NUMA node(s)", "On-line CPU(s) list")}
    except Exception as exc:  # noqa: BLE001
        res["lscpu"] = str(exc)
    res["affinity"] = len(os.sched_getaffinity(0))
    p = torch.cuda.get_device_properties(0)
    res["device"] = {"name": p.name, "cus": p.multi_processor_count, "mem_GB": round(p.total_memory / 2**30, 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
