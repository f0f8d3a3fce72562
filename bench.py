#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched k-NN hot path on MI355X.

Metric (BASELINE.json): Mqueries/s, knn = 1, 3-D float32, metric_l2_squared,
7 733 372-point tree / 7 200 863 queries (BASELINE configs[1]).  One "step" is one
pass of the whole query batch through ``ptk_search_knn_device`` (device-side
Morton ordering + traversal kernel [+ RCCL gather of the (index, distance) pairs
to rank 0 when N > 1]).  Tree points and queries are resident in HBM before the
timed region; the tree is built and uploaded once, outside it.

    python bench.py --gpus N --steps K --warmup W

For N > 1 the driver launches this file under ``torch.distributed.run``, one rank per GPU,
tree replicated on every GPU, queries independent (no data-path collective).  Default
``--scaling strong`` = BASELINE configs[3] literally: ONE 7.2 M batch cut into N contiguous
shards, rank 0 collecting every rank's (index, distance) rows with one RCCL gather per step,
issued asynchronously so that it overlaps the next step's search (all gathers complete inside the
timed region); ``value`` is that.  The same run then times the weak form (every rank searches a
full 7.2 M batch of its own: same cloud, another seed) and reports it as the extra object
``"weak"``.  ``--scaling weak`` makes the weak form ``value`` instead.

At N = 1 the line also carries (none of it inside the timed region of ``value``):
``host_buffers`` (the host-pointer entry ``ptk_search_knn`` on pageable arrays: H2D + search +
D2H), ``also`` (the same search on Morton-ordered queries and on cloud U), ``config3`` (knn = 16
and the radius search of BASELINE configs[2] with their own roofline figures), ``pipelined``
(two batches in flight on two HIP streams).

Rank 0 prints ONE JSON line; see README.md / DESIGN.md for the field meanings.
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
#: The k = 1 traversal is phase 1, the capped phase 2 and the cooperative search of what the cap left
#: (+ the usually empty redo pass); the algorithmic bytes cover all of them.
# The traversal kernels of the k = 1 search, by the prefixes of their names in a rocprofv3 kernel trace, and those
# names as `rocprofv3 --kernel-trace --stats` prints them for the default launch (profiles/r04*_stats.txt).
TRAVERSAL_KERNELS = ("ptk::knn1_phase1", "ptk::knn1_phase2", "ptk::knn1_coop", "ptk::knn1_redo")
TRAVERSAL_KERNEL_NAMES = ("ptk::knn1_phase1u_kernel<4, ptk::MetricL2>", "ptk::knn1_phase2_kernel<12, 64, 4, ptk::MetricL2>",
                          "ptk::knn1_coop_kernel<32, 96, true, true, ptk::MetricL2>",
                          "ptk::knn1_coop_kernel<16, 96, false, false, ptk::MetricL2>",
                          "ptk::knn1_redo_kernel<16, 64, 4, ptk::MetricL2>")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cloud", choices=["L", "U"], default="L",
                    help="L = LiDAR-like room scan (headline), U = uniform cube")
    ap.add_argument("--order", choices=["generated", "morton"], default="generated",
                    help="order in which the caller hands over the queries")
    ap.add_argument("--k", type=int, default=1)
    ap.add_argument("--n", type=int, default=None, help="tree points (default: config 2)")
    ap.add_argument("--nq", type=int, default=None, help="queries (default: config 2)")
    ap.add_argument("--points", default=None,
                    help="tree points from a file instead of a synthetic cloud: the reference's write_bin format "
                         "(float32 triples, e.g. the Bremen scans0.bin), .fvecs / .bvecs or .npy")
    ap.add_argument("--queries", default=None, help="query points from a file (same formats)")
    ap.add_argument("--leaf", type=int, default=10)
    ap.add_argument("--reorder", choices=["auto", "on", "off"], default="auto")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="N > 1: which form is `value`: strong = one batch cut into N shards (BASELINE configs[3]), "
                         "weak = a full batch per GPU; the other one is reported as an extra object")
    ap.add_argument("--no-extras", action="store_true",
                    help="N = 1: skip host_buffers / also / config3 (the headline line only)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams successive steps are issued on round-robin (single GPU only).  1 (default): one "
                         "batch at a time, the per-kernel durations are those of an isolated call.  > 1: batches "
                         "overlap (the handle keeps one scratch block per stream); kernel_ms then covers "
                         "kernels that shared the GPU")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the two-streams context measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--single-process-worker", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seconds", type=float, default=16.0,
                    help="approximate budget of the CPU baseline samples (OpenMP and single thread, two query orders)")
    ap.add_argument("--counter-sample", type=int, default=200_000,
                    help="queries sampled for the algorithmic-bytes visit counters")
    return ap.parse_args()


def algorithmic_bytes(ref, q_sample, k, dim):
    """Mean algorithmic bytes per query, SURVEY.md 8(d):
    B = 4*dim + 8*k + 16*N_branch + 8*N_leaf + (4*dim + 4)*N_pts,
    with the visit counts of the reference traversal (from the oracle)."""
    _, cnt = ref.search_knn(q_sample, k, counters=True)
    mean = cnt.astype(np.float64).mean(axis=0)
    b = 4 * dim + 8 * k + 16 * mean[0] + 8 * mean[1] + (4 * dim + 4) * mean[2]
    return float(b), {"n_branch": float(mean[0]), "n_leaf": float(mean[1]), "n_pts": float(mean[2])}


def measured_traffic_c3(prefixes):
    """The same for the kernels of BASELINE configs[2] (profiles/*_c3_traffic.json, tools/pmc_cmd.sh on
    tools/bench_config3.py): GB per launch of the kernels whose names start with one of `prefixes`, and the file."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_c3_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            kernels = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None, None
    # (a step may launch a kernel more than once -- the two capped launches of a large k > 1 batch --: bytes per STEP,
    # the kernel launched least often among the matched ones being the one a step launches once)
    hit = [rec for name, rec in kernels.items()
           if any(name.startswith(p) for p in prefixes) and "hbm_bytes_per_dispatch" in rec]
    steps = min((rec.get("dispatches_FETCH_SIZE", 1) for rec in hit), default=1) or 1
    total = sum(rec["hbm_bytes_per_dispatch"] * rec.get("dispatches_FETCH_SIZE", steps) / steps for rec in hit)
    return (round(total / 1e9, 3), os.path.basename(files[-1])) if total else (None, None)


def measured_traffic(prefixes):
    """HBM bytes per launch of the traversal kernels from the newest committed
    ``profiles/*_traffic.json`` (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
    command, tools/pmc_run.sh; 2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes).
    Counters cannot be collected from inside the process, hence the file; None if absent."""
    import glob

    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))
                   if "_c3_" not in os.path.basename(f) and "_forest" not in os.path.basename(f))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            kernels = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None, None
    total, seen = 0.0, 0
    for name, rec in kernels.items():
        if any(name.startswith(p) for p in prefixes) and "hbm_bytes_per_dispatch" in rec:
            total += rec["hbm_bytes_per_dispatch"]
            seen += 1
    return (total, os.path.basename(files[-1])) if seen else (None, None)


def host_cpus():
    """(model name, sockets, physical cores this process may run on, logical CPUs it may run on)."""
    allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set(range(os.cpu_count() or 1))
    model, cores, sockets = "unknown", set(), set()
    try:
        cpu, phys, core = None, None, None
        with open("/proc/cpuinfo") as f:
            for line in f.read().splitlines() + [""]:
                if not line.strip():
                    if cpu is not None and cpu in allowed:
                        cores.add((phys, core if core is not None else cpu))
                        sockets.add(phys)
                    cpu, phys, core = None, None, None
                    continue
                key, _, val = line.partition(":")
                key, val = key.strip(), val.strip()
                if key == "processor":
                    cpu = int(val)
                elif key == "physical id":
                    phys = int(val)
                elif key == "core id":
                    core = int(val)
                elif key == "model name":
                    model = val
    except (OSError, ValueError):
        pass
    n_cores = len(cores) if cores else len(allowed)
    return model, max(len(sockets), 1), n_cores, len(allowed)


def cpu_baseline_subprocess(args):
    """The CPU baseline runs in a process of its own: one OpenMP thread per PHYSICAL core, pinned
    (OMP_PLACES=cores OMP_PROC_BIND=spread).  An OpenMP runtime that binds pins the thread that loads it as well --
    here that would be the thread that drives the GPU and whose affinity the library's build threads inherit."""
    import subprocess

    env = dict(os.environ)
    env["OMP_PLACES"] = "cores"
    env["OMP_PROC_BIND"] = "spread"
    env.pop("OMP_NUM_THREADS", None)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--cloud", args.cloud, "--order", args.order,
           "--k", str(args.k), "--leaf", str(args.leaf), "--cpu-seconds", str(args.cpu_seconds)]
    for name in ("n", "nq", "points", "queries"):
        v = getattr(args, name)
        if v is not None:
            cmd += [f"--{name}", str(v)]
    try:
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, check=True)
        return json.loads(out.stdout.decode().strip().splitlines()[-1])
    except (subprocess.SubprocessError, ValueError, IndexError) as exc:
        log(f"[bench] cpu baseline failed: {exc!r}")
        return None


def interleave_memory():
    """Spreads every page this process touches from now on round-robin over the NUMA nodes with memory (what
    `numactl --interleave=all` does; the boxes have no numactl, so the set_mempolicy system call directly).  The
    reference builds its tree on one thread: first touch would put all of it -- and the points, and the queries -- on that
    thread's node, and the OpenMP threads of the other socket would search it across the socket link, faster or
    slower by whichever socket the build thread happened to start on.  Returns a description for the bench line."""
    import ctypes
    try:
        nodes = sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit())
        with_memory = []
        for nd in nodes:
            try:
                with open(f"/sys/devices/system/node/node{nd}/meminfo") as f:
                    total = [ln for ln in f if "MemTotal" in ln]
                if total and int(total[0].split()[-2]) > 0:
                    with_memory.append(nd)
            except OSError:
                pass
        if len(with_memory) < 2:
            return f"one NUMA node with memory ({len(nodes)} listed): nothing to interleave"
        mask_words = max(with_memory) // 64 + 1
        mask = (ctypes.c_ulong * mask_words)()
        for nd in with_memory:
            mask[nd // 64] |= 1 << (nd % 64)
        libc = ctypes.CDLL(None, use_errno=True)
        MPOL_INTERLEAVE, SYS_set_mempolicy = 3, 238  # x86-64
        if libc.syscall(SYS_set_mempolicy, MPOL_INTERLEAVE, ctypes.byref(mask), mask_words * 64 + 1) != 0:
            return f"set_mempolicy(MPOL_INTERLEAVE) failed (errno {ctypes.get_errno()}): first-touch placement"
        return f"pages interleaved over NUMA nodes {with_memory} (set_mempolicy(MPOL_INTERLEAVE), as numactl --interleave=all)"
    except (OSError, ValueError) as exc:
        return f"NUMA policy left alone ({exc!r})"


def cpu_baseline_worker(args):
    model_cores = host_cpus()  # before any OpenMP runtime is loaded and binds this thread
    numa = interleave_memory()  # before the clouds and the tree are allocated
    from pico_tree_amd import datasets as ds

    if args.points:
        pts, q = ds.load_points(args.points), ds.load_points(args.queries)
    else:
        pts, q = ds.config2_clouds(args.cloud, args.n or ds.CONFIG2_N, args.nq or ds.CONFIG2_NQ)
    if args.order == "morton":
        q = np.ascontiguousarray(q[ds.morton_order(q)])
    print(json.dumps(cpu_baseline(pts, q, args.k, args.leaf, args.cpu_seconds, model_cores, numa)), flush=True)


def cpu_baseline(pts, q, k, leaf, seconds, cpus=None, numa=None):
    """Times the reference (oracle/_ref, compiled from the reference's own headers) or, failing that, the oracle
    port, on the host cores, on bounded samples: its OpenMP schedule(dynamic,128) loop on one pinned thread per
    physical core, and one thread alone; on the queries as given to the GPU and on a Morton-sorted copy (order alone
    moves a CPU kd-tree by an order of magnitude).  Every figure: one untimed warm-up pass over
    a slice, then passes over fresh slices until the time budget is used; the MEDIAN and the FASTEST pass are both
    reported, with the host's load average when the leg began (the boxes are shared: four GPU slots per host)."""
    import oracle
    from pico_tree_amd import datasets as ds

    kind = "reference" if oracle.have_reference() else "port"
    t0 = time.perf_counter()
    cpu = oracle.Oracle(pts, leaf, kind)
    build_s = time.perf_counter() - t0
    model, sockets, cores, logical = cpus if cpus is not None else host_cpus()
    cores = max(1, cores)
    nsorted = min(len(q), 2_000_000)
    q_sorted = np.ascontiguousarray(q[:nsorted][ds.morton_order(q[:nsorted])])

    def loadavg():
        try:
            return [round(x, 1) for x in os.getloadavg()]  # other tenants of the host show up here
        except OSError:
            return None

    legs = {}

    def rate(name, queries, threads, chunk, budget, min_passes=3):
        cpu.set_threads(threads)
        chunk = max(1, min(chunk, len(queries) // (min_passes + 1)))
        load = loadavg()
        cpu.search_knn(queries[:chunk], k)  # warm-up: pages touched, threads started
        rates, at, used = [], chunk, 0.0
        while (used < budget or len(rates) < min_passes) and at + chunk <= len(queries):
            t0 = time.perf_counter()
            cpu.search_knn(queries[at:at + chunk], k)
            dt = time.perf_counter() - t0
            rates.append(chunk / dt / 1e6)
            used += dt
            at += chunk
        rates.sort()
        q1, q3 = (float(np.percentile(rates, p)) for p in (25, 75))
        legs[name] = {"fastest": round(rates[-1], 4), "median": round(float(np.median(rates)), 4),
                      "slowest": round(rates[0], 4), "q1": round(q1, 4), "q3": round(q3, 4), "iqr": round(q3 - q1, 4),
                      "passes": len(rates), "queries_per_pass": chunk, "threads": threads, "host_loadavg": load}
        return legs[name]

    # (passes of the all-cores legs long enough -- tens of milliseconds -- that one descheduled thread does not decide
    # them; every pass takes a FRESH slice of the batch, so nothing is served from a cache warmed by the pass before)
    # The as-given all-cores leg: at least five passes, the MEDIAN is the figure (the fastest pass
    # of three did not repeat across runs -- 77.5 / 31.4 / 15.6 Mq/s on one CPU model).  The hosts are shared (four GPU
    # slots each): when the load average says other tenants hold more than a quarter of the cores, `value` falls back
    # to the median of a 32-thread leg, which other tenants disturb least, and says so.
    omp = rate("value", q, cores, 900_000, seconds * 0.35, min_passes=5)
    few = min(32, cores)
    omp32 = rate("threads32_value", q, few, 450_000, seconds * 0.15, min_passes=5)
    omp_sorted = rate("morton_sorted_queries_value", q_sorted, cores, 300_000, seconds * 0.15, min_passes=5)
    one = rate("single_thread_value", q, 1, 50_000, max(1.0, seconds * 0.2))
    one_sorted = rate("single_thread_morton_sorted_value", q_sorted, 1, 100_000, max(1.0, seconds * 0.15))
    cpu.close()
    load0 = omp["host_loadavg"][0] if omp["host_loadavg"] else 0.0
    loaded = load0 > cores / 4.0
    # (the figure is the reference's BETTER leg: on the two-socket hosts 32 spread threads beat all 128 cores on the
    # as-given, incoherent queries -- 21.5 against 9.0 Mq/s -- because the all-cores run is bound by the caches and the
    # socket link; on a loaded host the 32-thread leg is also the one other tenants disturb least)
    main = omp32 if (loaded or omp32["median"] > omp["median"]) else omp
    loaded = loaded or main is omp32
    return {"value": main["median"], "value_iqr": main["iqr"], "value_fastest": main["fastest"], "unit": "Mqueries/s",
            "cores": few if main is omp32 else cores, "kind": kind,
            "host_loaded": bool(load0 > cores / 4.0), "host_loadavg": omp["host_loadavg"],
            "all_cores_value": omp["median"], "all_cores_iqr": omp["iqr"], "all_cores_threads": cores,
            "threads32_value": omp32["median"], "threads32_iqr": omp32["iqr"],
            "cpu": model, "sockets": sockets, "logical_cpus": logical,
            "threads": f"{few if main is omp32 else cores} of {cores} physical cores (OMP_PLACES={os.environ.get('OMP_PLACES')}, "
                       f"OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')})",
            "numa": numa,
            "sample": f"OpenMP schedule(dynamic,128), queries in the order given to the GPU: warm-up + {main['passes']} passes of "
                      f"{main['queries_per_pass']} fresh queries each on {main['threads']} threads; `value` = MEDIAN pass, "
                      f"`value_iqr` = interquartile range of the passes"
                      + (f" -- the {few}-thread leg is the figure (the faster of the two, or the host was loaded: load average "
                         f"{load0}, {cores} cores); the all-cores leg ({omp['median']} Mq/s, IQR {omp['iqr']}) is reported beside it"
                         if main is omp32 else ""),
            "morton_sorted_queries_value": omp_sorted["median"],
            "morton_sorted_queries_value_iqr": omp_sorted["iqr"],
            "morton_sorted_queries_value_fastest": omp_sorted["fastest"],
            "morton_sorted_queries_sample": f"the first {nsorted} queries Morton-sorted, warm-up + {omp_sorted['passes']} passes "
                                            f"of {omp_sorted['queries_per_pass']} fresh queries; median pass",
            "single_thread_value": one["median"], "single_thread_value_fastest": one["fastest"],
            "single_thread_sample": f"warm-up + {one['passes']} passes of {one['queries_per_pass']} queries as given; median pass",
            "single_thread_morton_sorted_value": one_sorted["median"],
            "single_thread_morton_sorted_value_fastest": one_sorted["fastest"],
            "single_thread_morton_sorted_sample": f"warm-up + {one_sorted['passes']} passes of {one_sorted['queries_per_pass']} sorted queries; median pass",
            "build_s": round(build_s, 3),
            "legs": legs,
            "note": "every figure is the MEDIAN of at least five passes over fresh slices (r03-r05 reported the fastest of "
                    "three, which did not repeat on the shared hosts: 77.5 / 31.4 / 15.6 Mq/s); the as-given all-cores "
                    "figure is cache- and NUMA-bound (incoherent queries over a 163 MB tree), the Morton-sorted one is the "
                    "reference's best case and the one the GPU/CPU ratio of DESIGN.md section 8 is quoted against"}


def time_device_knn(tree, dq, k, steps, warmup=2):
    """(ms per step, profile dict, last output) of `steps` device-resident searches on the current stream."""
    import torch

    out = torch.empty((dq.shape[0], k, 2), dtype=torch.int32, device=dq.device)
    for _ in range(warmup):
        tree.search_knn(dq, k, out)
    torch.cuda.synchronize()
    tree.profile(enable=True, reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        tree.search_knn(dq, k, out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    prof = tree.profile(enable=False, reset=True)
    return ms, prof, out


def roofline_of(b_per_q, nq, kernel_ms):
    achieved = (b_per_q * nq) / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    return {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "kernel_ms": round(kernel_ms, 4),
            "bytes_per_query": round(b_per_q, 1)}


def also_entry(pt, ds, oracle, cloud, order, pts, q, tree, leaf, steps, sample):
    """The headline search on another (cloud, query order): Mq/s, kernel time, roofline fraction, parity sample."""
    import torch

    if order == "morton":
        q = np.ascontiguousarray(q[ds.morton_order(q)])
    dq = torch.from_numpy(q).to(f"cuda:{tree.info()['device']}")
    ms, prof, out = time_device_knn(tree, dq, 1, steps)
    ref = oracle.Oracle(pts, leaf, "port")
    rng = np.random.default_rng(7)
    cs = np.sort(rng.choice(len(q), size=min(sample, len(q)), replace=False))
    want, cnt = ref.search_knn(q[cs], 1, counters=True)
    ref.close()
    mean = cnt.astype(np.float64).mean(axis=0)
    b = 12 + 8 + 16 * mean[0] + 8 * mean[1] + 16 * mean[2]
    got = pt.DeviceNeighbors(out).numpy()[cs][:, None]
    kernel_ms = prof["search_ms"] / max(int(prof["launches"]), 1)
    return {"cloud": cloud, "query_order": order, "value": round(len(q) / ms / 1e3, 3), "unit": "Mqueries/s",
            "ms_per_step": round(ms, 4), "steps": steps, "reorder_ms": round(prof["reorder_ms"] / steps, 4),
            "parity_sample_ok": bool(got.tobytes() == want.tobytes()),
            "roofline": roofline_of(b, len(q), kernel_ms)}


def approximate_entry(pt, oracle, pts, q, tree, dq, leaf, e, steps, sample):
    """The headline search as an approximate one, search_knn(pts, 1, e) (search_visitor.hpp:165-193): every candidate
    distance divided by e before it is compared, so the answer depends on the visit order and phase 2 runs every
    continuation to its end in the reference's order -- no cap, no cooperative tail."""
    import torch

    out = torch.empty((len(q), 1, 2), dtype=torch.int32, device=dq.device)
    tree.search_knn(dq, 1, e, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tree.search_knn(dq, 1, e, out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    ref = oracle.Oracle(pts, leaf, "port")
    cs = np.sort(np.random.default_rng(7).choice(len(q), size=min(sample, len(q)), replace=False))
    want = ref.search_knn(q[cs], 1, e=e)
    ref.close()
    got = pt.DeviceNeighbors(out).numpy()[cs][:, None]
    return {"e": e, "value": round(len(q) / ms / 1e3, 3), "unit": "Mqueries/s", "ms_per_step": round(ms, 4), "steps": steps,
            "parity_sample_ok": bool(got.tobytes() == want.tobytes())}


def metrics_and_f64_entries(pt, oracle, pts, q, leaf, device, steps, sample):
    """The headline clouds under the other metrics of the reference's Python module (metric_l1, metric_lpinf:
    ``ptk_tree_set_metric``) and over float64 points (``ptk_tree64_*``): knn = 1 and knn = 16, device-resident, each with a
    parity sample against the oracle under that metric / dtype.  metric_l1 takes the two-phase k = 1 search and the capped
    k > 1 search since r06; metric_lpinf cannot and runs the general kernel."""
    import torch

    nq = len(q)
    cs = np.sort(np.random.default_rng(11).choice(nq, size=min(sample, nq), replace=False))
    res = {"metrics": {}, "f64": {}}

    def time_knn(tree, dq, k, rows):
        out = torch.empty((dq.shape[0], k, 2), dtype=rows, device=dq.device)
        for _ in range(2):
            tree.search_knn(dq, k, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tree.search_knn(dq, k, out)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, out

    dq = torch.from_numpy(q).to(f"cuda:{device}")
    for metric in ("L1", "LPInf"):
        tree = pt.KdTree(pts, pt.Metric[metric], leaf, device=device)
        ref = oracle.Oracle(pts, leaf, "port", metric)
        ref.set_threads(ref.max_threads())
        entry = {}
        for k in (1, 16):
            ms, out = time_knn(tree, dq, k, torch.int32)
            got = pt.DeviceNeighbors(out).numpy()[cs]
            want = ref.search_knn(q[cs], k)
            entry[f"knn{k}"] = {"value": round(nq / ms / 1e3, 1), "unit": "Mqueries/s", "ms_per_step": round(ms, 4),
                                "parity_sample_ok": bool(got.reshape(want.shape).tobytes() == want.tobytes())}
            del out
        res["metrics"][metric] = entry
        ref.close()
        tree.close()
    del dq
    p64, q64 = pts.astype(np.float64), q.astype(np.float64)
    tree = pt.KdTree(p64, pt.Metric.L2Squared, leaf, device=device)
    ref = oracle.Oracle(p64, leaf, "port", dtype=np.float64)
    ref.set_threads(ref.max_threads())
    dq = torch.from_numpy(q64).to(f"cuda:{device}")
    for k in (1, 16):
        ms, out = time_knn(tree, dq, k, torch.int64)
        got = pt.DeviceNeighbors(out).numpy()[cs]
        want = ref.search_knn(q64[cs], k)
        ok = np.array_equal(got["index"].reshape(want["index"].shape), want["index"]) and \
            np.ascontiguousarray(got["distance"]).tobytes() == np.ascontiguousarray(want["distance"]).tobytes()
        coop = tree.knn_coop_counts()  # (the capped launch + cooperative finish of ptk_kernels_coop64.hpp: 0 = it ran uncapped)
        res["f64"][f"knn{k}"] = {"value": round(nq / ms / 1e3, 1), "unit": "Mqueries/s", "ms_per_step": round(ms, 4),
                                 "parity_sample_ok": bool(ok), "handed_over": coop["cooperative"], "redone": coop["redone"]}
        del out
        # every 48th query: a batch the size of a piece of a host-buffer call (before r06 any double batch took as long
        # as the longest search of the cloud -- 2.9 / 4.3 ms here)
        dqs = dq[::48].contiguous()
        ms_s, out_s = time_knn(tree, dqs, k, torch.int64)
        got_s = pt.DeviceNeighbors(out_s).numpy()
        want_s = ref.search_knn(q64[::48], k)
        ok_s = np.array_equal(got_s["index"].reshape(want_s["index"].shape), want_s["index"]) and \
            np.ascontiguousarray(got_s["distance"]).tobytes() == np.ascontiguousarray(want_s["distance"]).tobytes()
        res["f64"][f"knn{k}_150k"] = {"queries": int(dqs.shape[0]), "ms_per_step": round(ms_s, 4), "rows_equal": bool(ok_s),
                                      "handed_over": tree.knn_coop_counts()["cooperative"]}
        del out_s, dqs
    # the double radius search on 20 000 queries through the host entry (both passes + the copies: there is no device-buffer
    # form in double); capped since r06 (any such call used to take twice the cloud's longest search: 6.7 ms here)
    qs = np.ascontiguousarray(q64[:: max(1, nq // 20_000)][:20_000])
    tree.search_radius(qs, 1.0)
    t0 = time.perf_counter()
    for _ in range(3):
        got_r = tree.search_radius(qs, 1.0)
    ms_r = (time.perf_counter() - t0) / 3 * 1e3
    want_off, want_flat = ref.search_radius(qs, 1.0)
    ok_r = np.array_equal(got_r.offsets, want_off) and np.array_equal(got_r.flat["index"], want_flat["index"]) and \
        np.ascontiguousarray(got_r.flat["distance"]).tobytes() == np.ascontiguousarray(want_flat["distance"]).tobytes()
    res["f64"]["radius_20k_host"] = {"queries": int(len(qs)), "radius_squared": 1.0, "ms_per_call": round(ms_r, 3), "hits": int(want_off[-1]),
                                     "rows_equal": bool(ok_r), "handed_over": tree.knn_coop_counts()["cooperative"]}
    ref.close()
    tree.close()
    return res


def config3_entries(pt, oracle, pts, q, tree, dq, leaf, sample):
    """BASELINE configs[2] on the headline clouds: knn = 16 and search_radius r = 1.0 (squared radius 1.0)."""
    import torch

    nq = len(q)
    ref = oracle.Oracle(pts, leaf, "port")
    ref.set_threads(ref.max_threads())
    rng = np.random.default_rng(7)
    cs = np.sort(rng.choice(nq, size=min(sample, nq), replace=False))
    res = {}
    # ---- knn = 16: one kernel (k-list in registers)
    k, steps = 16, 5
    ms, prof, out = time_device_knn(tree, dq, k, steps, warmup=1)
    want, cnt = ref.search_knn(q[cs], k, counters=True)
    got = pt.DeviceNeighbors(out).numpy()[cs]
    del out
    mean = cnt.astype(np.float64).mean(axis=0)
    b = 12 + 8 * k + 16 * mean[0] + 8 * mean[1] + 16 * mean[2]
    kernel_ms = prof["search_ms"] / max(int(prof["launches"]), 1)
    r = roofline_of(b, nq, kernel_ms)
    r["kernel"] = ("ptk::knn_reg_kernel<16, 16, 64, 64, 5, ptk::MetricL2, true> (capped; two launches side by side, the front "
                   "fifth of the launch order on a second stream) + ptk::knn_coop_kernel<16, 128> behind each + "
                   "ptk::knn_redo_kernel<16, ...>: the launches of a step, timed together from the first to the last")
    r["traffic"], r["traffic_source"] = measured_traffic_c3(["ptk::knn_reg_kernel<16,", "ptk::knn_coop_kernel<16,",
                                                             "ptk::knn_redo_kernel<16,"])
    r["traffic_static"] = True  # (GB per launch, 2 x FETCH_SIZE + WRITE_SIZE of committed rocprofv3 --pmc passes, as above)
    r["visits_per_query"] = {"n_branch": round(mean[0], 2), "n_leaf": round(mean[1], 2), "n_pts": round(mean[2], 2)}
    try:
        coop = tree.knn_coop_counts()  # queries finished by a wavefront each / redone by one lane, and why
    except Exception as exc:  # noqa: BLE001
        coop = str(exc)
    res["knn16"] = {"value": round(nq / ms / 1e3, 3), "unit": "Mqueries/s", "ms_per_step": round(ms, 4), "steps": steps,
                    "parity_sample_ok": bool(got.tobytes() == want.tobytes()), "roofline": r,
                    "long_searches": coop}
    # ---- knn = 4 and knn = 8 through the same three launches (what capping the long searches buys is largest there)
    for kk in (4, 8):
        ms_k, prof_k, out_k = time_device_knn(tree, dq, kk, 5, warmup=1)
        want_k = ref.search_knn(q[cs[:20_000]], kk)
        got_k = pt.DeviceNeighbors(out_k).numpy()[cs[:20_000]]
        del out_k
        res[f"knn{kk}"] = {"value": round(nq / ms_k / 1e3, 3), "unit": "Mqueries/s", "ms_per_step": round(ms_k, 4),
                           "kernel_ms": round(prof_k["search_ms"] / max(int(prof_k["launches"]), 1), 4),
                           "parity_sample_ok": bool(got_k.tobytes() == want_k.tobytes())}
    # ---- knn = 16 on one shard of configs[3] (the first eighth of the batch): a capped launch ends with the lanes that
    # ran to their cap, so the cap follows the batch (knn_cap of ptk_backend.hip; at a fixed 256 this step took 1.9 ms)
    per = (nq + 7) // 8
    ms_s, prof_s, out_s = time_device_knn(tree, dq[:per], 16, 10, warmup=2)
    head = cs[cs < per][:20_000]
    got_s = pt.DeviceNeighbors(out_s).numpy()[head]
    del out_s
    try:
        coop_s = tree.knn_coop_counts()
    except Exception as exc:  # noqa: BLE001
        coop_s = str(exc)
    res["knn16_shard_of_8"] = {"queries": per, "value": round(per / ms_s / 1e3, 3), "unit": "Mqueries/s",
                               "ms_per_step": round(ms_s, 4),
                               "kernel_ms": round(prof_s["search_ms"] / max(int(prof_s["launches"]), 1), 4),
                               "parity_sample_ok": bool(got_s.tobytes() == ref.search_knn(q[head], 16).tobytes()),
                               "long_searches": coop_s}
    # ---- radius: count pass that lists the leaves with hits + scan + fill pass that replays the lists
    radius, steps = 1.0, 3
    for _ in range(2):  # (the 6 GB of rows are a block of torch's allocator from the second call on)
        off, raw = tree.search_radius_device(dq, radius)
        torch.cuda.synchronize()
        del off, raw
    tree.profile(enable=True, reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        off = raw = None  # (the rows of the step before go back to the allocator before the next are asked for)
        off, raw = tree.search_radius_device(dq, radius)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    prof = tree.profile(enable=False, reset=True)
    hits = int(off[-1].item())
    rs = cs[:: max(1, len(cs) // 20_000)]
    o2, flat, rcnt = ref.search_radius(q[rs], radius, counters=True)
    offs = off.cpu().numpy()
    rawn = raw.cpu().numpy()
    ok = True
    for j in range(0, len(rs), 37):
        a = rawn[offs[rs[j]]:offs[rs[j] + 1]].view(pt.NEIGHBOR)[:, 0]
        ok = ok and a.tobytes() == flat[int(o2[j]):int(o2[j + 1])].tobytes()
    import hashlib
    small = {"shard_of_8": per, "20k": 20_000}
    prefix = {name: (int(offs[n]), hashlib.sha256(rawn[:int(offs[n])].tobytes()).hexdigest()) for name, n in small.items()}
    del raw, rawn
    mean = rcnt.astype(np.float64).mean(axis=0)
    b = 12 + 8 * (hits / nq) + 16 * mean[0] + 8 * mean[1] + 16 * mean[2]
    kernel_ms = prof["search_ms"] / steps
    r = roofline_of(b, nq, kernel_ms)
    r["kernel"] = "ptk::radius_list_kernel<16, 64, 5, ptk::MetricL2> + ptk::radius_replay_kernel<8, 32, ptk::MetricL2>"
    r["traffic"], r["traffic_source"] = measured_traffic_c3(["ptk::radius_list_kernel", "ptk::radius_replay_kernel"])
    r["traffic_static"] = True
    r["visits_per_query"] = {"n_branch": round(mean[0], 2), "n_leaf": round(mean[1], 2), "n_pts": round(mean[2], 2)}
    res["radius"] = {"radius_squared": radius, "value": round(nq / ms / 1e3, 3), "unit": "Mqueries/s",
                     "ms_per_step": round(ms, 4), "steps": steps, "hits_per_query": round(hits / nq, 2),
                     "parity_sample_ok": bool(ok), "roofline": r}
    # ---- the radius search on one shard of configs[3] and on 20 k queries: the list pass capped, the long queries counted
    # and filled by a wavefront each (ptk_kernels_coopr.hpp; before, ANY batch of this cloud took 2.2 ms -- its longest query)
    for name, n in small.items():
        d = dq[:n]
        for _ in range(2):
            off_s, raw_s = tree.search_radius_device(d, radius)
        torch.cuda.synchronize()
        same = (int(off_s[-1].item()), hashlib.sha256(raw_s.cpu().numpy().tobytes()).hexdigest()) == prefix[name]
        tree.profile(enable=True, reset=True)
        t0 = time.perf_counter()
        for _ in range(10):
            off_s = raw_s = None
            off_s, raw_s = tree.search_radius_device(d, radius)
        torch.cuda.synchronize()
        ms_s = (time.perf_counter() - t0) / 10 * 1e3
        prof_s = tree.profile(enable=False, reset=True)
        try:
            coop_r = tree.radius_coop_counts()
        except Exception as exc:  # noqa: BLE001
            coop_r = str(exc)
        res[f"radius_{name}"] = {"queries": n, "ms_per_step": round(ms_s, 4), "kernel_ms": round(prof_s["search_ms"] / 10, 4),
                                 "value": round(n / ms_s / 1e3, 3), "unit": "Mqueries/s", "rows_equal_full_batch": bool(same),
                                 "long_searches": coop_r}
        del off_s, raw_s
    ref.close()
    return res


def shard_entry(pt, oracle, pts, q, tree, leaf, parts, b_per_q_full):
    """One shard of BASELINE configs[3] on THIS GPU: the first ceil(nq / parts) rows of the batch against the whole
    tree -- what each of `parts` GPUs does per step before the gather.  Step time with the profiling events off
    (steps back to back), kernel split from a second, profiled pass; rows byte-equal to the full-batch rows."""
    import torch

    nq = len(q)
    per = (nq + parts - 1) // parts
    dq = torch.from_numpy(np.ascontiguousarray(q[:per])).to(f"cuda:{tree.info()['device']}")
    out = torch.empty((per, 1, 2), dtype=torch.int32, device=dq.device)
    steps = 50
    for _ in range(5):
        tree.search_knn(dq, 1, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tree.search_knn(dq, 1, out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    tree.profile(enable=True, reset=True)
    for _ in range(steps):
        tree.search_knn(dq, 1, out)
    torch.cuda.synchronize()
    prof = tree.profile(enable=False, reset=True)
    counts = tree.knn1_counts()
    ref = oracle.Oracle(pts, leaf, "port")
    sample = np.linspace(0, per - 1, num=min(4096, per), dtype=np.int64)
    want = ref.search_knn(q[:per][sample], 1)
    ref.close()
    got = pt.DeviceNeighbors(out).numpy()
    kernel_ms = prof["search_ms"] / steps
    return {"what": f"one shard of BASELINE configs[3] on one GPU: rows [0, {per}) of the batch, the whole tree",
            "parts": parts, "queries": per, "ms_per_step": round(ms, 4), "steps": steps,
            "value": round(per / ms / 1e3, 3), "unit": "Mqueries/s",
            "projected_parts_x_value": round(parts * per / ms / 1e3, 1),
            "kernels_ms": {"reorder": round(prof["reorder_ms"] / steps, 4),
                           "phase1": round((prof["search_ms"] - prof["search_tail_ms"]) / steps, 4),
                           "class_order": round(prof["other_ms"] / steps, 4),
                           "phase2_cooperative_replay": round(prof["search_tail_ms"] / steps, 4),
                           "note": "HIP events inside the library, a second pass of the same steps"},
            "counts": counts, "parity_sample_ok": bool(got[sample][:, None].tobytes() == want.tobytes()),
            "rows": got, "roofline": roofline_of(b_per_q_full, per, kernel_ms)}


def quantised_entry(pt, ds, oracle, pts, q, leaf, grid, device, steps, sample, shift=0.0):
    """The headline search with every coordinate of points and queries snapped to a grid (what a scan stored with a
    few decimals looks like: exact ties between distances, coincident points -- at a grid of 1.0 piles of hundreds of
    them, which the reference's builder peels apart one level per point); shift: the queries moved that part of a
    cell off the grid along every axis, so that none of them sits on a pile."""
    import torch

    p2 = np.ascontiguousarray(np.round(pts / grid) * grid, dtype=np.float32)
    q2 = np.ascontiguousarray(np.round(q / grid) * grid + np.float32(shift * grid), dtype=np.float32)
    tree = pt.KdTree(p2, pt.Metric.L2Squared, leaf, device=device)
    what = f"L, coordinates snapped to a grid of {grid}" + (f", queries {shift} of a cell off the grid" if shift else "")
    e = also_entry(pt, ds, oracle, what, "generated", p2, q2, tree, leaf, steps, sample)
    e["tree_depth"] = int(tree.info()["max_depth"])
    e["piles"] = tree.piles()
    e["counts"] = tree.knn1_counts()
    return e


def forest_counter_fraction(ms_per_launch):
    """HBM bytes per launch of ptk::forest_knn_kernel from the newest committed profiles/*_forest_pmc.txt (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE, tools/pmc_forest.sh; 2 x FETCH_SIZE + WRITE_SIZE, KiB units) over `ms_per_launch` and the
    HBM peak; None without such a file."""
    import glob
    import re

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_forest_pmc.txt")))
    if not files:
        return None
    fetch = write = None
    try:
        with open(files[-1]) as f:
            for line in f:
                if "forest_knn_kernel" not in line:
                    continue
                m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)", line)
                if m and m.group(1) == "FETCH_SIZE":
                    fetch = float(m.group(4)) * 1024.0
                elif m:
                    write = float(m.group(4)) * 1024.0
    except OSError:
        return None
    if fetch is None:
        return None
    gb = (2.0 * fetch + (write or 0.0)) / 1e9
    return {"hbm_gb_per_launch": round(gb, 2), "frac": round(gb / (ms_per_launch * 1e-3) / HBM_PEAK_GBS, 4),
            "source": os.path.basename(files[-1]), "static": True}


def config5_entry(pt, ds, device):
    """BASELINE configs[4]: approximate knn = 10 through the kd-forest (8 trees, leaf 32, 64 leaves per tree) on a
    SIFT-1M-shaped synthetic cloud; recall against brute force on the GPU (tools/bench_forest.py)."""
    import torch

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_forest import exact_knn, recall

    n, nq, dim, trees, leaf, leaves, k, steps = 1_000_000, 10_000, 128, 8, 32, 64, 10, 10
    pts = ds.sift_like_cloud(n, dim, seed=1)
    q = ds.sift_like_cloud(nq, dim, seed=2)
    t0 = time.perf_counter()
    forest = pt.KdForest(pts, leaf, trees, seed=1, device=device)
    build_s = time.perf_counter() - t0
    dq = torch.from_numpy(q).to(f"cuda:{device}")
    res = forest.search_knn(dq, k, leaves)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = forest.search_knn(dq, k, leaves)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    got = res.numpy()["index"].reshape(nq, -1)
    exact = exact_knn(torch.from_numpy(pts).to(dq.device), dq, k)
    row_bytes = trees * leaves * leaf * dim * 4  # every leaf row a query reads
    achieved = row_bytes * nq / (ms * 1e-3) / 1e9
    return {"what": f"kd_forest approximate knn={k}: {trees} trees, max_leaf_size {leaf}, {leaves} leaves per tree, "
                    f"{n} x {dim} float32 (synthetic SIFT-like mixture), {nq} queries per batch",
            "value": round(nq / ms * 1e3, 1), "unit": "queries/s", "ms_per_batch": round(ms, 3), "steps": steps,
            "recall_at_1": round(recall(got, exact, 1), 4), f"recall_at_{k}": round(recall(got, exact, k), 4),
            "recall_reference": "brute force on the GPU (float32 expansion, candidates re-ranked in float64)",
            "queue_entries_dropped": int(forest.dropped), "host_build_upload_s": round(build_s, 2),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "kernel": "ptk::forest_knn_kernel",
                         "bytes_per_query": row_bytes,
                         "frac_by_counters": forest_counter_fraction(ms),
                         "note": "`frac` prices the ALGORITHMIC leaf rows (every visited leaf's points, 512 B each); "
                                 "`frac_by_counters` is what the HBM counters saw per launch (2 x FETCH_SIZE + WRITE_SIZE of "
                                 "the newest committed profiles/*_forest_pmc.txt) over this run's launch time: the "
                                 "difference is what the Infinity Cache and the L2s serve"}}


def single_process_worker(args):
    """The other form of BASELINE configs[3]: ONE process, the tree replicated on the first
    `--single-process-worker` devices of the node, the batch on devices[0], its ranges and their rows moved over xGMI
    by ptk_multi_search_knn_device (grouped RCCL send / recv inside libptk).  Run as a child of rank 0 with a time
    limit, beside the idle ranks of the main run: whatever happens here cannot take the main line with it."""
    import torch

    import oracle
    import pico_tree_amd as pt
    from pico_tree_amd import datasets as ds

    ndev = args.single_process_worker
    pts, q = ds.config2_clouds(args.cloud, args.n or ds.CONFIG2_N, args.nq or ds.CONFIG2_NQ)
    multi = pt.MultiKdTree(pts, args.leaf, devices=list(range(ndev)))
    torch.cuda.set_device(0)
    dq = torch.from_numpy(q).cuda()
    for _ in range(max(1, args.warmup)):
        out = multi.search_knn(dq, args.k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = multi.search_knn(dq, args.k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    sample = np.linspace(0, len(q) - 1, num=4096, dtype=np.int64)
    ref = oracle.Oracle(pts, args.leaf, "port")
    want = ref.search_knn(q[sample], args.k)
    ref.close()
    got = out.numpy()[sample]
    got = got if args.k > 1 else got[:, None]
    print(json.dumps({"devices": ndev, "value": round(len(q) / ms / 1e3, 3), "unit": "Mqueries/s",
                      "ms_per_step": round(ms, 4), "steps": args.steps, "queries_per_step": len(q),
                      "what": "one process, pt.MultiKdTree: queries and rows on devices[0], ranges over xGMI "
                              "(ptk_multi_search_knn_device)",
                      "parity_sample_ok": bool(got.tobytes() == want.tobytes())}), flush=True)
    return 0


def run_single_process_form(args, ndev, timeout_s=240):
    """Rank 0: the single-process form as a child with a time limit; a dict either way."""
    cmd = [sys.executable, os.path.abspath(__file__), "--single-process-worker", str(ndev), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--cloud", args.cloud, "--k", str(args.k), "--leaf", str(args.leaf)]
    if args.n:
        cmd += ["--n", str(args.n)]
    if args.nq:
        cmd += ["--nq", str(args.nq)]
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "GROUP_RANK", "ROLE_RANK",
                        "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    try:
        done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
        lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
        if done.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"devices": ndev, "error": (done.stderr or done.stdout)[-400:]}
    except subprocess.TimeoutExpired:
        return {"devices": ndev, "error": f"no result within {timeout_s} s"}
    except Exception as exc:  # noqa: BLE001 -- context only, never the line
        return {"devices": ndev, "error": repr(exc)[:400]}


def main():
    args = parse_args()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args)
    if args.single_process_worker:
        return single_process_worker(args)
    import torch
    import torch.distributed as dist

    import pico_tree_amd as pt
    from pico_tree_amd import datasets as ds

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    n_gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pt.warmup(local_rank)  # (libptk starts loading its code object for THIS rank's device now, beside the generation of the clouds)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    n = args.n or ds.CONFIG2_N
    nq = args.nq or ds.CONFIG2_NQ
    k = args.k
    dim = 3

    def own_batch(r):  # rank r's batch of the weak form: same cloud, its own seed
        if r == 0:
            return None
        qq = (ds.lidar_cloud(nq, seed=2 + r, pose=(3.0, 1.5), unit_scale=20.0) if args.cloud == "L"
              else ds.uniform_cloud(nq, 3, seed=2 + r, scale=100.0))
        return np.ascontiguousarray(qq[ds.morton_order(qq)]) if args.order == "morton" else qq

    t0 = time.perf_counter()
    if args.points or args.queries:
        if not (args.points and args.queries):
            raise SystemExit("--points and --queries go together")
        pts, q = ds.load_points(args.points), ds.load_points(args.queries)
        if pts.shape[1] != 3 or q.shape[1] != 3:
            raise SystemExit("bench.py times the 3-D search; use tools/bench_forest.py / tools/bench_nd.py otherwise")
        n, nq = len(pts), len(q)
        args.n, args.nq = n, nq  # not the BASELINE sizes: the metric string and the extras follow
    else:
        pts, q = ds.config2_clouds(args.cloud, n, nq)
    if args.order == "morton":
        q = np.ascontiguousarray(q[ds.morton_order(q)])
    gen_s = time.perf_counter() - t0

    t0 = time.perf_counter()
    tree = pt.KdTree(pts, pt.Metric.L2Squared, args.leaf, device=local_rank)
    build_s = time.perf_counter() - t0
    create = {"first_s": round(build_s, 3), "first_phases": {k_: round(v, 3) for k_, v in tree.create_phases().items()}}
    tree.set_reorder({"auto": pt.REORDER_AUTO, "on": pt.REORDER_ON, "off": pt.REORDER_OFF}[args.reorder])
    info = tree.info()
    if rank == 0:
        log(f"[bench] clouds {args.cloud}/{args.order}: gen {gen_s:.1f}s, host build+upload {build_s:.1f}s, "
            f"nodes {info['n_nodes']}, depth {info['max_depth']}, HBM {info['device_bytes'] / 1e6:.1f} MB")

    from pico_tree_amd.sharded import ShardedSearch, padded_shard, shard_of

    def fence(sharded):
        sharded.finish()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_form(weak):
        """Times `args.steps` steps of one form.  weak: every rank its own full batch; otherwise one batch
        in `world` contiguous shards (at world == 1 the two coincide).  Returns a dict."""
        if weak and world > 1:  # every rank: its whole batch
            sh = shard_of(nq * world, world, rank)
            per, lo, hi = nq, 0, nq
            mine = own_batch(rank)
            dq = torch.from_numpy(q if mine is None else mine).to(dev)
        else:     # contiguous ranges of ceil(nq / world) rows of ONE batch, in caller order
            sh = shard_of(nq, world, rank)
            per, lo, hi = sh.per, sh.lo, sh.hi
            dq = torch.from_numpy(padded_shard(q, sh)).to(dev)
        total = nq * world if (weak and world > 1) else nq
        sharded = ShardedSearch(sh, lambda qq, oo: tree.search_knn(qq, k, oo).raw,
                                lambda: torch.empty((per, k, 2), dtype=torch.int32, device=dev))
        n_streams = max(1, args.streams) if world == 1 else 1
        if n_streams > 1:  # successive batches on several streams of ONE handle (per-stream scratch blocks)
            streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
            lanes = [ShardedSearch(sh, lambda qq, oo: tree.search_knn(qq, k, oo).raw,
                                   lambda: torch.empty((per, k, 2), dtype=torch.int32, device=dev), depth=1)
                     for _ in range(n_streams)]
            issued = [0]

            def step():
                i = issued[0] % n_streams
                issued[0] += 1
                with torch.cuda.stream(streams[i]):
                    return lanes[i].step(dq)
        else:
            def step():
                return sharded.step(dq)

        for _ in range(args.warmup):
            step()
        fence(sharded)
        tree.profile(enable=True, reset=True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        fence(sharded)
        elapsed = time.perf_counter() - t0
        prof = tree.profile(enable=False, reset=True)
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        return {"weak": weak and world > 1, "elapsed": elapsed, "prof": prof, "out": out, "dq": dq, "sharded": sharded,
                "per": per, "lo": lo, "hi": hi, "total": total, "n_streams": n_streams,
                "ms_per_step": elapsed / args.steps * 1e3, "value": total / (elapsed / args.steps) / 1e6}

    main_is_weak = args.scaling == "weak" and world > 1
    form = run_form(main_is_weak)
    other = run_form(not main_is_weak) if world > 1 else None
    weak = form["weak"]
    per, lo, hi, out, dq, sharded, prof = (form[x] for x in ("per", "lo", "hi", "out", "dq", "sharded", "prof"))
    total_queries, n_streams = form["total"], form["n_streams"]
    ms_per_step, value = form["ms_per_step"], form["value"]

    # Context, never `value`: the same batches with two in flight on two HIP streams of the same handle
    # ("Pipelined batches").  Single GPU, default stream count only.
    pipelined = None
    if world == 1 and n_streams == 1 and not args.no_pipelined:
        two = [torch.cuda.Stream(device=dev) for _ in range(2)]
        outs2 = [torch.empty((per, k, 2), dtype=torch.int32, device=dev) for _ in range(2)]

        def run2(steps):
            for i in range(steps):
                with torch.cuda.stream(two[i % 2]):
                    tree.search_knn(dq, k, outs2[i % 2])
            torch.cuda.synchronize()
        run2(2 * max(1, args.warmup))
        t0 = time.perf_counter()
        run2(args.steps)
        dt = time.perf_counter() - t0
        pipelined = {"streams": 2, "value": round(total_queries * args.steps / dt / 1e6, 3), "unit": "Mqueries/s",
                     "ms_per_step": round(dt / args.steps * 1e3, 4),
                     "rows_equal_single_stream": bool(torch.equal(outs2[(args.steps - 1) % 2], out))}
        del outs2

    result = None
    if rank == 0:
        import oracle

        # Sanity: the timed output equals the oracle on a sample (never timed).
        res = pt.DeviceNeighbors(out).numpy()
        sample = np.linspace(0, (hi - lo) - 1, num=min(4096, hi - lo), dtype=np.int64)
        ref_small = oracle.Oracle(pts, args.leaf, "port")
        want = ref_small.search_knn(q[lo:hi][sample], k)
        got = res[sample] if k > 1 else res[sample][:, None]
        parity_ok = bool(got.tobytes() == want.tobytes())
        if world > 1:  # and the rows rank 0 gathered from rank 1 (last step)
            rows = sharded.result(rows_per_rank=per if weak else None)
            if weak:
                q1 = own_batch(1)
                got1 = pt.DeviceNeighbors(rows[per:2 * per]).numpy()[sample]
                want1 = ref_small.search_knn(q1[sample], k)
            else:
                sh1 = shard_of(nq, world, 1)
                s1 = np.linspace(sh1.lo, sh1.hi - 1, num=min(4096, sh1.rows), dtype=np.int64)
                got1 = pt.DeviceNeighbors(rows).numpy()[s1]
                want1 = ref_small.search_knn(q[s1], k)
            got1 = got1 if k > 1 else got1[:, None]
            parity_ok = parity_ok and bool(got1.tobytes() == want1.tobytes())

        rng = np.random.default_rng(7)
        cs = rng.choice(nq, size=min(args.counter_sample, nq), replace=False)
        b_per_q, visits = algorithmic_bytes(ref_small, q[np.sort(cs)], k, dim)
        ref_small.close()

        launches = max(int(prof["launches"]), 1)
        kernel_ms = prof["search_ms"] / launches
        q_per_launch = prof["queries"] / launches if prof["launches"] else per
        achieved = (b_per_q * q_per_launch) / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic, traffic_src = (None, None)
        if k == 1 and not args.n and not args.nq and args.cloud == "L" and world == 1:
            traffic, traffic_src = measured_traffic(TRAVERSAL_KERNELS)
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": None if traffic is None else round(traffic / 1e9, 3),
                    "traffic_unit": "GB per launch (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc)",
                    "traffic_static": True,  # counters cannot be read inside the process: per-launch figure of the
                    "traffic_source": traffic_src,  # committed rocprofv3 --pmc passes of this same command (that file)
                    "algorithmic_gb_per_launch": round(b_per_q * q_per_launch / 1e9, 3),
                    "kernel": " + ".join(TRAVERSAL_KERNEL_NAMES) if k == 1 else "ptk::knn_reg_kernel / ptk::knn_kernel",
                    "kernel_ms": round(kernel_ms, 4), "reorder_ms": round(prof["reorder_ms"] / launches, 4),
                    "other_ms": round(prof["other_ms"] / launches, 4),
                    "bytes_per_query": round(b_per_q, 1), "queries_per_launch": int(q_per_launch),
                    "visits_per_query": {kk: round(v, 2) for kk, v in visits.items()}}

        extras = {}
        if world == 1 and not args.no_extras and k == 1 and not args.n and not args.nq:
            # (a) the host-pointer entry on pageable arrays: H2D + search + D2H (SURVEY 8d asks for both figures)
            hsteps = 5
            warm = [tree.search_knn(q, 1) for _ in range(3)]  # (the wrapper's pool of page-locked blocks fills: two
            del warm                                           # result arrays are alive at a time in the loop below)
            t0 = time.perf_counter()
            for _ in range(hsteps):
                host_rows = tree.search_knn(q, 1)
            hdt = (time.perf_counter() - t0) / hsteps
            extras["host_buffers"] = {"value": round(nq / hdt / 1e6, 3), "unit": "Mqueries/s",
                                      "ms_per_step": round(hdt * 1e3, 4), "steps": hsteps,
                                      "what": "rows = tree.search_knn(q, 1) on a pageable numpy query array: 86 MB up, "
                                              "search, 58 MB down; a new result array every call, as search_knn(pts, k) "
                                              "of the reference's module returns one (def_kd_tree.cpp:73-82) -- built on "
                                              "a page-locked block of the wrapper's pool that the device writes directly",
                                      "rows_equal_device_run": bool(host_rows.tobytes() == res.tobytes())}
            # the overload that fills the caller's array (search_knn(pts, k, nns), def_kd_tree.cpp): a PAGEABLE numpy
            # array of the caller's own here: the rows are staged through the handle's pinned ring
            own_rows = np.empty(nq, dtype=pt.NEIGHBOR)
            tree.search_knn(q, 1, own_rows)
            t0 = time.perf_counter()
            for _ in range(hsteps):
                tree.search_knn(q, 1, own_rows)
            hdt2 = (time.perf_counter() - t0) / hsteps
            extras["host_buffers"]["result_array_handed_in"] = {
                "value": round(nq / hdt2 / 1e6, 3), "ms_per_step": round(hdt2 * 1e3, 4),
                "what": "search_knn(q, 1, nns) with nns a pageable numpy array of the caller",
                "rows_equal_device_run": bool(own_rows.tobytes() == res.tobytes())}
            del host_rows, own_rows
            # queries kept in page-locked memory as well (pt.empty_pinned): no staging in either direction
            q_pin = pt.empty_pinned(q.shape, q.dtype)
            q_pin[...] = q
            pin_rows = tree.search_knn(q_pin, 1)
            t0 = time.perf_counter()
            for _ in range(hsteps):
                pin_rows = tree.search_knn(q_pin, 1)
            hdt3 = (time.perf_counter() - t0) / hsteps
            extras["host_buffers"]["queries_page_locked_too"] = {
                "value": round(nq / hdt3 / 1e6, 3), "ms_per_step": round(hdt3 * 1e3, 4),
                "rows_equal_device_run": bool(pin_rows.tobytes() == res.tobytes())}
            del pin_rows
            # ... and the caller's OWN pageable arrays page-locked in place for the duration (ptk_host_register): what an
            # application with long-lived buffers does instead of allocating from ptk_host_alloc
            nns_reg = np.empty(nq, dtype=pt.NEIGHBOR)
            with pt.registered(q), pt.registered(nns_reg):
                tree.search_knn(q, 1, nns_reg)
                t0 = time.perf_counter()
                for _ in range(hsteps):
                    tree.search_knn(q, 1, nns_reg)
                hdt4 = (time.perf_counter() - t0) / hsteps
            extras["host_buffers"]["callers_arrays_registered"] = {
                "value": round(nq / hdt4 / 1e6, 3), "ms_per_step": round(hdt4 * 1e3, 4),
                "what": "search_knn(q, 1, nns) with the caller's numpy arrays page-locked in place (pt.registered)",
                "rows_equal_device_run": bool(nns_reg.tobytes() == res.tobytes())}
            del nns_reg
            # knn = 16 through the same entry (0.92 GB of rows down)
            warm = [tree.search_knn(q, 16) for _ in range(2)]
            del warm
            t0 = time.perf_counter()
            for _ in range(3):
                rows16 = tree.search_knn(q, 16)
            hdt16 = (time.perf_counter() - t0) / 3
            extras["host_buffers"]["knn16"] = {"value": round(nq / hdt16 / 1e6, 3), "ms_per_step": round(hdt16 * 1e3, 3),
                                               "what": "rows = tree.search_knn(q, 16): 86 MB up, 922 MB of rows down"}
            del rows16, q_pin
            # (a2) one shard of configs[3] (strong scaling decided on one GPU) and its rows against the full batch
            sh8 = shard_entry(pt, oracle, pts, q, tree, args.leaf, 8, b_per_q)
            sh8["rows_equal_full_batch"] = bool(sh8.pop("rows").tobytes() == res[:sh8["queries"]].tobytes())
            extras["shard_of_8"] = sh8
            # (a3) what a second creation of the same handle takes (the first pays for loading the code object)
            t0 = time.perf_counter()
            tree_b = pt.KdTree(pts, pt.Metric.L2Squared, args.leaf, device=local_rank)
            create["steady_s"] = round(time.perf_counter() - t0, 3)
            create["steady_phases"] = {k_: round(v, 3) for k_, v in tree_b.create_phases().items()}
            create["what"] = ("KdTree(points): build (partitions of the top levels on the device, the subtrees below by the "
                              "host threads of the library; phase host_build_s), re-encoding for the device, upload + "
                              "point gather; first = first handle of the process (the code object loads from "
                              "pt.warmup(device) on, beside the generation of the clouds)")
            del tree_b
            extras["create"] = create
            # (b) config 3 on the same clouds
            if args.cloud == "L" and args.order == "generated":
                extras["config3"] = config3_entries(pt, oracle, pts, q, tree, dq, args.leaf, 100_000)
            if k == 1:
                extras["approximate"] = approximate_entry(pt, oracle, pts, q, tree, dq, args.leaf, 1.05, 10, 20_000)
            if args.cloud == "L" and args.order == "generated":  # (the other metrics and float64 on the headline clouds)
                try:
                    extras.update(metrics_and_f64_entries(pt, oracle, pts, q, args.leaf, local_rank, 5, 20_000))
                except Exception as exc:  # noqa: BLE001 -- a leg that fails says so instead of taking the line with it
                    extras["metrics"] = {"error": repr(exc)}
            # (c) the headline search on the other query order and on the other cloud
            also = []
            other_order = "morton" if args.order == "generated" else "generated"
            also.append(also_entry(pt, ds, oracle, args.cloud, other_order, pts, q, tree, args.leaf, 10, 100_000))
            oc = "U" if args.cloud == "L" else "L"
            del dq
            pts2, q2 = ds.config2_clouds(oc, n, nq)
            tree2 = pt.KdTree(pts2, pt.Metric.L2Squared, args.leaf, device=local_rank)
            for order in ("generated", "morton"):
                also.append(also_entry(pt, ds, oracle, oc, order, pts2, q2, tree2, args.leaf, 10, 100_000))
            del tree2, pts2, q2
            if args.cloud == "L":  # tie-prone data: every coordinate a multiple of 0.1
                also.append(quantised_entry(pt, ds, oracle, pts, q, args.leaf, 0.1, local_rank, 10, 100_000))
                # ... and of 1.0: piles of up to 600 coincident points (the k = 1 view without them, ptk_piles.hpp)
                also.append(quantised_entry(pt, ds, oracle, pts, q, args.leaf, 1.0, local_rank, 10, 100_000))
                also.append(quantised_entry(pt, ds, oracle, pts, q, args.leaf, 1.0, local_rank, 10, 100_000, shift=0.3))
            extras["also"] = also
            # (d) BASELINE configs[4]
            extras["config5"] = config5_entry(pt, ds, local_rank)

        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline_subprocess(args)

        def parallelism(w):
            return ((f"{world} x {nq} queries (one full batch per GPU), tree replicated" if w
                     else f"one batch of {nq} queries cut into {world} shards, tree replicated")
                    + (", (index, distance) rows gathered on rank 0 over RCCL, overlapped with the next step"
                       if world > 1 else ""))

        result = {
            "metric": "Mqueries/sec, knn=1 3D L2, 7.73M-pt tree / 7.20M queries" if k == 1 and not args.n
                      else f"Mqueries/sec, knn={k} 3D L2",
            "value": round(value, 3), "unit": "Mqueries/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not args.points else f"files {os.path.basename(args.points)} / {os.path.basename(args.queries)}",
            "config": {"workload": f"BASELINE configs[{3 if world > 1 and not weak else 1}]: cloud {args.cloud} "
                                   f"({'LiDAR-like room scan' if args.cloud == 'L' else 'uniform cube'}), "
                                   f"{n} tree points / {nq} queries, knn={k}, max_leaf_size={args.leaf}, "
                                   f"sliding midpoint",
                       "query_order": args.order, "reorder": args.reorder, "streams": n_streams,
                       "parallelism": parallelism(weak),
                       "queries_per_step": int(total_queries),
                       "tree_nodes": int(info["n_nodes"]), "tree_depth": int(info["max_depth"]),
                       "host_build_upload_s": round(build_s, 2)},
            "parity_sample_ok": parity_ok,
            "roofline": roofline,
            "pipelined": pipelined,
            "cpu_baseline": cpu,
        }
        if other is not None:  # the other form of the same run (N > 1)
            result["weak" if other["weak"] else "strong"] = {
                "value": round(other["value"], 3), "unit": "Mqueries/s", "ms_per_step": round(other["ms_per_step"], 4),
                "queries_per_step": int(other["total"]), "parallelism": parallelism(other["weak"])}
        if world > 1 and pt.device_count() >= world and not args.no_extras:
            # (the other ranks wait for the key below on the host, their devices idle: a collective barrier would
            # keep a polling kernel on every one of them while the child measures)
            result["single_process"] = run_single_process_form(args, world)
        result.update(extras)
        print(json.dumps(result), flush=True)
    if world > 1:
        if not args.no_extras:
            try:
                from datetime import timedelta

                store = dist.distributed_c10d._get_default_store()
                if rank == 0:
                    store.set("ptk_bench_single_process_done", "1")
                else:
                    store.wait(["ptk_bench_single_process_done"], timedelta(seconds=400))
            except Exception:  # noqa: BLE001 -- the barrier below orders the ranks either way
                pass
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and result is not None and not result["parity_sample_ok"]:
        raise SystemExit("parity check failed")


if __name__ == "__main__":
    main()
