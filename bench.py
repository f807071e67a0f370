"""Benchmark of the BoostClassifier.fit() hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One *step* is one complete ``BoostClassifier(**kw).fit(X)`` from a host scipy CSR matrix: input validation
(dd.py:149-160), PCIe upload, HVG prologue, ``n_iters`` boosting iterations and the result gather are all inside
the timed region.  Metric: cells/s = cells * steps / wall-clock, the maximum over ranks, whole job.  The same fit on
counts that ``clf.stage(X)`` made resident beforehand is timed separately and reported as ``value_resident``
(N = 1 only).  With N > 1 the boosting iterations are sharded over the ranks (one process per GPU, RCCL all-gather
of the per-iteration result rows), so total work is fixed: "strong".  ``python bench.py --gpus N`` without a
torchrun environment launches the N ranks itself.

Rank 0 prints ONE JSON line.  Besides the driver contract it carries
  roofline      -- the dominant GPU kernel of the timed region: algorithmic bytes (or flops) per launch
                   divided by its average launch duration measured with HIP events on the library's stream;
  cpu_baseline  -- the CPU oracle (a port of the reference's numpy/scipy/sklearn path) timed on this host
                   on a bounded sample of the same workload (N=1, rank 0 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced
# reads, + WRITE_SIZE) of this command at the headline workload: profiles/pmc_traffic.json, written by
# `profiles/summarise_pmc.py --traffic-json` at the end of profiles/collect.sh (no literals here: re-collect, copy, commit)
def _pmc_traffic():
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            d = json.load(fh)
        return d.get("kernels", {}), d.get("source")
    except Exception:
        return {}, None


PMC_TRAFFIC_GB, PMC_TRAFFIC_SOURCE = _pmc_traffic()


def _cpu_full():
    """The full-size CPU run (`bench.py --cpu-full`: two iterations of the oracle on ALL cells, minutes) is too long for the
    default run; the committed record of the last one is quoted beside the sampled baseline."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cpu_full.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as fh:
            cb = json.load(fh)["cpu_baseline"]
        return {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"],
                "source": os.path.relpath(files[-1], ROOT)}
    except Exception:
        return None


def cpu_allowance():
    """CPUs' worth of time the container may use (cgroup CFS quota), or None when there is no limit.  A pod can show 256 CPUs and be
    allowed 16: threads beyond the allowance are throttled, so the CPU baseline is run on -- and reported with -- the allowance."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1, int(int(quota) / int(period)))
    except Exception:
        pass
    try:
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0:
            return max(1, quota // period)
    except Exception:
        pass
    return None


HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
LDS_PEAK_GBS = 256 * 128 * 2.4     # 256 CUs x 128 bytes per clock x 2.4 GHz = 78.6 TB/s of LDS reads (MI355X_MICROARCH.md)
FP64_PEAK_TFLOPS = 78.6      # FP64 vector / FP64 MFMA peak, dense
FP32_PEAK_TFLOPS = 157.3
BF16_PEAK_TFLOPS = 2500.0     # dense bf16 MFMA peak (MI355X_MICROARCH.md; not the 2:1-sparsity figure)
I8_PEAK_TOPS = 5000.0         # dense int8 MFMA: twice the bf16 rate (MI355X_MICROARCH.md: 16x16x64 / 32x32x32 i8 at 2x bf16; microbenchmark ceiling 4450, profiles/r05_bitplane_notes.txt)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cells", type=int, default=100_000)
    ap.add_argument("--genes", type=int, default=30_000)
    ap.add_argument("--density", type=float, default=0.03)
    ap.add_argument("--iters", type=int, default=10, help="n_iters (reference default 10)")
    ap.add_argument("--algorithm", default="phenograph")
    ap.add_argument("--scaling", action="store_true", help="standard_scaling=True")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-cells", type=int, default=20000)
    ap.add_argument("--cpu-full", action="store_true", help="cpu_baseline on ALL cells (two iterations of the oracle at the full size: "
                                                            "minutes and tens of GB of host memory; SURVEY.md section 8 d)")
    ap.add_argument("--instrumented-steps", type=int, default=2, help="extra fits with per-kernel HIP events (N=1) behind the kernel tables")
    ap.add_argument("--resident-steps", type=int, default=5, help="extra fits on staged counts for value_resident (N=1)")
    ap.add_argument("--no-exclusive", action="store_true", help="skip the extra single-context fits behind roofline_exclusive")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE", help="ddx_set_option switch for every context (include/ddx.h), repeatable")
    return ap.parse_args()


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torchrun: launch the N ranks (one process per GPU) ourselves."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def kernel_models(N, G, H, S, nnz_aug, C, k, L=None, knn_window=1.0, bitplane=None):
    """Algorithmic work per launch of the kernels that can dominate.

    HBM-bound kernels: bytes that must cross HBM once (matrix entries 8 B each + dense operands in and out).
    MFMA-bound kernels: useful flops of the distance screen (2 flop per component and pair, padded C)."""
    M = N + S
    L = L or (C + 10)
    CP = 32 if C <= 32 else 64
    Mp = -(-M // 512) * 512
    nsamp_tiles = min(max(512, (Mp // 16) // 24), Mp // 16)      # tiles of the bound pass's subset (stage_knn: max(512, tiles / 24))
    # Bit-plane route (csrc/k_bitplane.hip): the stored entries equal to 1 are multiplied as bitmaps on the int8 matrix cores, the
    # sparse kernels walk the others.  The sparse kernels' algorithmic bytes are then those of the entries they walk; the matrix-core
    # kernels are priced by the int8 multiply-adds of the formulation (every (row, column) bit against L columns x ND digits).
    bp = bitplane if (bitplane and bitplane.get("active")) else None
    nnz_sparse = (bp["rest_original"] + bp["rest_synthetic"]) if bp else nnz_aug
    models = {
        # name: (bound, unit, work per launch, peak)
        "spmm_rows": ("hbm", "GB/s", (8 * nnz_sparse + 8 * M * L + 8 * H * L + 8 * (M + 1)) / 1e9, HBM_PEAK_GBS),
        "spmm_cols": ("hbm", "GB/s", (8 * nnz_sparse + 8 * H * L + 8 * M * L + 16 * (H + 1)) / 1e9, HBM_PEAK_GBS),
        # distance screen on the bfloat16 MFMA.  USEFUL work: 2 flop per component and screened pair (the emit pass only
        # screens the tile pairs its first-component window admits: measured fraction).  The kernel ISSUES three products
        # (hi*hi, hi*lo, lo*hi) per pair on components padded to CP: see issued_per_launch below.
        "knn_emit": ("mfma", "TFLOP/s", knn_window * 2.0 * M * M * C / 1e12, BF16_PEAK_TFLOPS),
        "knn_bound": ("mfma", "TFLOP/s", 2.0 * M * nsamp_tiles * 16 * C / 1e12, BF16_PEAK_TFLOPS),
        "pca_orth": ("hbm", "GB/s", (24 * M * L) / 1e9, HBM_PEAK_GBS),
        "doublet_fill": ("hbm", "GB/s", (8 * (nnz_aug * 2 * S / max(M + S, 1)) * 2) / 1e9, HBM_PEAK_GBS),
        "lognorm_rows": ("hbm", "GB/s", (8 * nnz_aug + 8 * M) / 1e9, HBM_PEAK_GBS),
        "lognorm_cols": ("hbm", "GB/s", (12 * nnz_aug) / 1e9, HBM_PEAK_GBS),
        # counting-sort mirror of the synthetic rows: columns read twice (4 B), raw values once (4 B), (row, raw) written once (8 B)
        "mirror_build": ("hbm", "GB/s", (20 * nnz_aug * 2 * S / max(M + S, 1)) / 1e9, HBM_PEAK_GBS),
    }
    issued = {"knn_emit": knn_window * 3 * 2.0 * Mp * Mp * CP / 1e12, "knn_bound": 3 * 2.0 * Mp * nsamp_tiles * 16 * CP / 1e12}
    if bp:
        nd = bp["digits"]
        nt32 = 4 if nd == 3 else 5                                # 32-wide tiles of the flattened (column, digit) index
        if bp.get("format") == "mx6":
            nd, nt32 = 6, 8                                       # six base-31 digits in FP6 (priced against the same int8 peak: option bp_format)
        elif bp.get("digits_early"):
            # sklearn's 7 power iterations + the projection = 16 products, 12 of them on the early digit count: a launch's average
            ne = bp["digits_early"]
            nd = (12 * ne + 4 * nd) / 16.0
            nt32 = (12 * (4 if ne == 3 else 3 if ne == 2 else 5) + 4 * nt32) / 16.0
        m_pad = -(-N // 256) * 256 + -(-S // 64) * 64             # rows the kernels touch (originals padded to 256, synthetic rows to 64)
        h_pad, k_rows = -(-H // 256) * 256, -(-N // 256) * 256 + -(-S // 256) * 256
        models["bitplane_rows"] = ("mfma", "TOP/s", 2.0 * M * H * L * nd / 1e12, I8_PEAK_TOPS)
        models["bitplane_cols"] = ("mfma", "TOP/s", 2.0 * M * H * L * nd / 1e12, I8_PEAK_TOPS)
        issued["bitplane_rows"] = 2.0 * m_pad * h_pad * nt32 * 32 / 1e12
        issued["bitplane_cols"] = 2.0 * (-(-H // 32) * 32) * k_rows * nt32 * 32 / 1e12
    return models, issued


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: every GPU needs its own rank")
    # DDX_BENCH_BACKEND=gloo (tests only): the ranks share the visible GPUs and the collectives run on the host, so that the
    # multi-rank control flow of this file can be exercised on a one-GPU box.  The driver's runs use nccl (= RCCL).
    backend = os.environ.get("DDX_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and backend == "nccl":
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {ndev} device(s) visible")
    device_index = local_rank % max(ndev, 1)
    if world > 1:
        torch.cuda.set_device(device_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{device_index}"))
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: {dist.get_world_size()} ranks joined, expected {args.gpus}")
    dev = f"cuda:{device_index}"

    from doubletdetection_amd import BoostClassifier, _lib
    from doubletdetection_amd._synthetic import make_counts

    for item in args.option:
        key, _, value = item.partition("=")
        _lib.OPTIONS[key] = value

    os.environ["DDX_TIMING"] = "0"              # the timed steps run the production path: no per-kernel HIP events
    t_gen = time.perf_counter()
    X = make_counts(args.cells, args.genes, density=args.density, device=dev, seed=20250227)
    t_gen = time.perf_counter() - t_gen
    torch.cuda.empty_cache()
    N, G = X.shape
    kw = dict(n_iters=args.iters, clustering_algorithm=args.algorithm, standard_scaling=args.scaling,
              random_state=0, n_jobs=-1, device=device_index)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_fit(resident=False, **extra):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            clf = BoostClassifier(**kw, **extra)
            if resident:
                clf.stage(X)                # validation + upload ahead of the clock (value_resident only)
            barrier()
            t0 = time.perf_counter()
            clf.fit(X)                      # headline: host CSR in, fitted classifier out (dd.py:135-214)
            barrier()
        return clf, time.perf_counter() - t0

    for _ in range(args.warmup):
        one_fit()
    elapsed = 0.0
    clf = None
    cpu0 = time.process_time()                   # CPU time of this rank's process (all its threads) over the timed steps
    for _ in range(args.steps):
        clf, dt = one_fit()
        elapsed += dt
    cpu_timed = time.process_time() - cpu0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        t = torch.tensor([cpu_timed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        cpu_timed = float(t.item())
    # ---- everything below is outside the timed region -------------------------------------------------------------------
    resident_elapsed = None
    exclusive = None
    timings = {}
    instr_elapsed, busy_ms = 0.0, 0.0
    diag = None                                  # an instrumented fit: it also reads the diagnostics the production path leaves on the device
    if world == 1 and args.resident_steps > 0:
        resident_elapsed = sum(one_fit(resident=True)[1] for _ in range(args.resident_steps))
    if args.instrumented_steps > 0:          # (every rank: a fit ends with the ranks' collective)
        # the same fits with two HIP events around every kernel scope (~1 400 scopes per fit): per-kernel times, and the
        # wall-clock share during which at least one scope was running (union of the scope intervals of all streams)
        os.environ["DDX_TIMING"] = "1"
        one_fit()
        for _ in range(args.instrumented_steps):
            c2, dt = one_fit()
            diag = c2
            instr_elapsed += dt
            busy_ms += c2._device_busy_ms or 0.0
            for name, (launches, ms) in c2._device_timings.items():
                a = timings.setdefault(name, [0, 0.0])
                a[0] += launches
                a[1] += ms
        if world == 1 and getattr(clf, "_lanes_used", 1) > 1 and not args.no_exclusive:      # (N = 1 only: a per-rank condition must not decide about fits that end in a collective)
            # and on a single device context: every kernel has the GPU to itself, so its HIP-event duration is the
            # kernel's own (with several contexts a launch is stretched by its neighbours)
            one_fit(streams_per_device=1)
            exclusive = one_fit(streams_per_device=1)[0]._device_timings
        os.environ["DDX_TIMING"] = "0"

    if rank == 0:
        H = clf._num_genes
        S = int(clf.boost_rate * N)
        nnz_aug = getattr(diag, "_last_nnz_aug", None) or getattr(clf, "_last_nnz_aug", None) or int(X.nnz * 1.0)
        C = clf.n_components
        L_ = C + 10
        k = 30 if args.algorithm == "phenograph" else 10
        bp_stats = getattr(clf, "_last_bitplane", None)
        models, issued = kernel_models(N, G, H, S, nnz_aug, C, k, knn_window=getattr(clf, "_last_knn_window", 1.0), bitplane=bp_stats)
        gpu_ms = {n: v[1] for n, v in timings.items() if v[0] > 0 and v[1] > 0}
        modelled = [n for n in gpu_ms if n in models]
        dominant = max(modelled, key=gpu_ms.get) if modelled else None      # the dominant kernel among those with a byte / flop model
        def roof(name, timings=timings):
            bound, unit, work, peak = models[name]
            avg_s = timings[name][1] / max(timings[name][0], 1) / 1e3
            achieved = work / avg_s
            out = {"bound": bound, "kernel": name, "achieved": round(achieved, 2), "peak": peak, "unit": unit,
                   "frac": round(achieved / peak, 4), "traffic": PMC_TRAFFIC_GB.get(name),
                   "traffic_source": PMC_TRAFFIC_SOURCE if name in PMC_TRAFFIC_GB else None,
                   "avg_launch_ms": round(avg_s * 1e3, 4), "launches": timings[name][0], "work_per_launch": round(work, 4)}
            if name in ("spmm_rows", "spmm_cols"):
                # the products are bound on chip: every stored entry reads one operand row (ld floats) from LDS -- how far the
                # launch is from THAT bound (the HBM fraction above says how far it is from the bound it could have)
                ent = nnz_aug if not (bp_stats and bp_stats.get("active")) else bp_stats["rest_original"] + bp_stats["rest_synthetic"]
                lds_gb = ent * (((L_ + 3) // 4) * 4) * 4 / 1e9
                out["lds_bound"] = {"bytes_per_launch_GB": round(lds_gb, 3), "peak_GBs": LDS_PEAK_GBS, "achieved_GBs": round(lds_gb / avg_s, 1),
                                    "frac": round(lds_gb / avg_s / LDS_PEAK_GBS, 4),
                                    "floor_ms": round(lds_gb / LDS_PEAK_GBS * 1e3, 3)}
            if name in issued:          # MFMA screens: useful flops above, what the kernel issues (3 products, padded) here
                out["issued_per_launch"] = round(issued[name], 4)
                out["issued_frac"] = round(issued[name] / avg_s / peak, 4)
            return out

        nsteps_i = max(args.instrumented_steps, 1)
        roofline_timed = roof(dominant) if dominant in models else None
        roofline_timed_all = [roof(n) for n in sorted(gpu_ms, key=gpu_ms.get, reverse=True) if n in models][:6]
        # When two device contexts share the GPU (the default), the HIP events around a launch also count the time the
        # launch spends queued behind / sharing CUs with the other stream's kernels: 1.1 ms for a product that runs
        # 0.66 ms alone, and rocprofv3's begin/end stamps (0.78 ms) agree with neither.  The roofline entry therefore
        # comes from the fit that follows the timed steps on ONE context (same process, same data, same kernels, each
        # alone on the GPU); it agrees with `rocprofv3 --kernel-trace --stats` of `DDX_STREAMS=1 bench.py`
        # (profiles/*_kernel_stats_1stream.csv).  The timed-region figures are kept beside it.
        roofline, roofline_all, roofline_source = roofline_timed, roofline_timed_all, "HIP events over the instrumented fits (several contexts share the GPU)"
        if exclusive:
            ex = {n: [v[0], v[1]] for n, v in exclusive.items()}
            order = [n for n in sorted(ex, key=lambda n: -ex[n][1]) if n in models]
            roofline, roofline_all = roof(order[0], ex), [roof(n, ex) for n in order[:6]]
            roofline_source = ("HIP events over one fit on a single device context, run right after the timed steps (in the "
                               "timed region several contexts share the GPU and a launch's events include queueing behind the "
                               "other streams: see roofline_shared_gpu)")
        if roofline:
            roofline["measured"] = roofline_source
        # one operator product as a whole against the bytes it would have to move as ONE pass over the stored entries (the figure
        # rounds 1-4 quoted for the single sparse kernel): all its kernels' time, bit-plane preparation included
        product = None
        src_t = ({n: [v[0], v[1]] for n, v in exclusive.items()} if exclusive else timings)
        if "spmm_rows" in src_t and "spmm_cols" in src_t:
            def per_launch(n):
                return src_t[n][1] / max(src_t[n][0], 1) if n in src_t else 0.0
            full_gb = (8 * nnz_aug + 8 * (N + S) * L_ + 8 * H * L_ + 8 * (N + S + 1)) / 1e9
            prep = per_launch("bitplane_prep")
            product = {}
            for side, mf in (("rows", "bitplane_rows"), ("cols", "bitplane_cols")):
                ms = per_launch("spmm_" + side) + per_launch(mf) + prep + (per_launch("spmm_sum") if side == "cols" else 0.0)
                product["A Q" if side == "rows" else "A^T Y"] = {
                    "ms_per_product": round(ms, 4), "kernels_ms": {"sparse": round(per_launch("spmm_" + side), 4), "matrix_cores": round(per_launch(mf), 4),
                                                                  "preparation": round(prep, 4)},
                    "one_pass_bytes_GB": round(full_gb, 4), "achieved_GBs": round(full_gb / (ms / 1e3), 1) if ms else None,
                    "frac_of_hbm_peak": round(full_gb / (ms / 1e3) / HBM_PEAK_GBS, 4) if ms else None}
        roofline_single = roofline
        if product and bp_stats and bp_stats.get("active"):
            # The dominant unit of work is one operator product, which this round is several launches: the bitmap product on the matrix
            # cores, the packed sparse kernel on the other entries, the partial sums and the operand preparation.  It is priced as a
            # whole against ONE pass over all stored entries (SURVEY section 8 d: 8 bytes per stored entry + the dense operand and
            # result), the slower side quoted; the kernels one by one follow in roofline_top_kernels.
            side = max(product, key=lambda k: product[k]["ms_per_product"])
            pr = product[side]
            names = ["bitplane_cols", "spmm_cols", "spmm_sum", "bitplane_prep"] if side == "A^T Y" else ["bitplane_rows", "spmm_rows", "bitplane_prep"]
            traffic = [PMC_TRAFFIC_GB.get(n) for n in names if n != "bitplane_prep"]
            roofline = {"bound": "hbm", "kernel": f"operator product {side} = " + " + ".join(names), "achieved": pr["achieved_GBs"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": pr["frac_of_hbm_peak"],
                        "traffic": round(sum(traffic), 4) if all(t is not None for t in traffic) else None,
                        "traffic_source": PMC_TRAFFIC_SOURCE, "avg_launch_ms": pr["ms_per_product"],
                        "launches": src_t["spmm_cols" if side == "A^T Y" else "spmm_rows"][0], "work_per_launch": pr["one_pass_bytes_GB"],
                        "parts_ms": pr["kernels_ms"], "measured": roofline_source,
                        "note": "achieved = bytes of one pass over all stored entries / the time of all launches of one product; the largest "
                                "single kernel is in roofline_dominant_kernel"}
        out = {
            "metric": "cells/sec for full BoostClassifier.fit() (default n_iters)",
            "value": round(N * args.steps / elapsed, 2),
            "timed_region": "BoostClassifier(**kw).fit(host scipy CSR): check_array-equivalent validation + PCIe upload + "
                            "HVG prologue + n_iters iterations + gather (dd.py:135-214); production path (no per-kernel events)",
            "value_resident": (round(N * args.resident_steps / resident_elapsed, 2) if resident_elapsed else None),
            "value_instrumented": (round(N * args.instrumented_steps / instr_elapsed, 2) if instr_elapsed else None),
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": ("int8 digits (30-bit fixed point) x bits on the matrix cores + f32 products, f64 sums" if (bp_stats and bp_stats.get("active"))
                      else "f32 products, f64 sums"),
            "data": "synthetic",
            "config": {"workload": f"synthetic {N}x{G} counts, {X.nnz / (N * G):.3%} nnz, n_iters={args.iters}, "
                                   f"n_top_var_genes=10000, n_components=30, boost_rate=0.25, "
                                   f"clustering_algorithm={args.algorithm}, standard_scaling={args.scaling}",
                       "n_iters": args.iters,
                       "sharding": f"iterations over {world} rank(s) x {getattr(clf, '_lanes_used', 1)} device context(s) "
                                   "(streams) per GPU",
                       "host_threads": os.cpu_count(), "host_cpu_allowance": cpu_allowance()},
            "roofline": roofline,
            "roofline_dominant_kernel": roofline_single,
            "roofline_top_kernels": roofline_all,
            "operator_product": product,
            "bitplane": bp_stats,
            "roofline_shared_gpu": roofline_timed,
            "roofline_shared_gpu_top_kernels": roofline_timed_all,
            "gpu_kernel_ms_per_step": {n: round(v / nsteps_i, 3) for n, v in sorted(gpu_ms.items(), key=lambda kv: -kv[1])},
            "gpu_kernel_ms_per_step_note": "HIP-event spans summed over all device contexts of the instrumented fits (overlapping "
                                           "streams: the sum exceeds the wall-clock); the timed steps themselves carry no events",
            "gpu_busy_frac": (round(busy_ms / 1e3 / instr_elapsed, 4) if instr_elapsed and busy_ms else None),
            "gpu_busy_frac_note": "union of the kernel-scope intervals of all streams / wall-clock of the instrumented fits",
            "host_seconds_last_step": {k2: round(v, 3) for k2, v in getattr(clf, "_host_timings", {}).items()},
            "host_cpu_seconds_per_step": round(cpu_timed / args.steps, 3),
            "host_cpu_seconds_per_step_note": "process CPU time (all threads) over the timed steps, summed over the ranks; upload form of the last fit: "
                                              + str(getattr(clf, "_upload_form_used", None)),
            "datagen_s": round(t_gen, 2),
            "notes": "PCA = sklearn's randomized SVD as 16 operator products per iteration (no dense H x H Gram is formed, DESIGN.md "
                     "section 3).  Since round 5 each product = the stored entries equal to 1 as bitmaps against 8-bit digits of the operand on "
                     "the int8 matrix cores (bitplane_rows / bitplane_cols: MFMA-bound rows of roofline_top_kernels, exact integer "
                     "arithmetic) + the other entries through the LDS-staged sparse kernel (spmm_rows / spmm_cols); `operator_product` "
                     "prices a whole product against one pass over all stored entries (an EFFECTIVE bandwidth of several launches: the "
                     "single-kernel rooflines are roofline_dominant_kernel / roofline_top_kernels).  Round 6: standard_scaling=True takes the same "
                     "route (1 / sd_j in the operand digits and the A^T Y epilogue; `bitplane.scaled`, `bitplane.demoted_columns`).  The kNN "
                     "distance screen runs on the bf16 MFMA (knn_emit / knn_bound)",
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(X, args, kw)
        full = _cpu_full()
        if full and (N, G) == (100_000, 30_000):
            out["cpu_baseline_full"] = full
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(X, args, kw):
    """The CPU oracle ("port" of the reference's path: scipy sparse ops, dense log matrix, sklearn
    randomized PCA, exact kNN, the same deterministic Louvain compiled for the host) on a bounded
    row sample of the benchmark matrix, all host cores available to BLAS / sklearn."""
    from doubletdetection_amd import _lib
    from oracle import dd_oracle as orc

    allowed = cpu_allowance()
    cores = min(os.cpu_count() or 1, allowed) if allowed else (os.cpu_count() or 1)
    n = X.shape[0] if args.cpu_full else min(args.cpu_sample_cells, X.shape[0])
    rows = np.sort(np.random.default_rng(0).choice(X.shape[0], size=n, replace=False))
    sample = X[rows]
    iters = 2        # ~10 s of CPU work on the GPU box's host

    def native_louvain(indptr, indices, weights, gamma, seed):
        return _lib.louvain(indptr, indices, weights, gamma, seed)[0].astype(np.int64)

    def native_best_of(indptr, indices, weights, gamma, seed, q_tol):
        return _lib.louvain_best_of(indptr, indices, weights, gamma, seed, q_tol, threads=min(20, cores))[0].astype(np.int64)

    def knn_allowed_cores(emb, k, include_self):
        # sklearn's exact search on the cores the container may use (n_jobs=-1 would start one job per VISIBLE CPU)
        from sklearn.neighbors import NearestNeighbors

        kk = k if include_self else k + 1
        nn = NearestNeighbors(n_neighbors=kk, algorithm="kd_tree" if not include_self else "brute", n_jobs=cores).fit(emb)
        dist, idx = nn.kneighbors(emb)
        if not include_self:
            idx, dist = idx[:, 1:], dist[:, 1:]
        return idx, dist

    okw = dict(n_iters=iters, clustering_algorithm=kw["clustering_algorithm"], standard_scaling=kw["standard_scaling"],
               random_state=0, louvain_fn=native_louvain, best_of_fn=native_best_of,
               knn_fn=knn_allowed_cores if allowed else orc._knn_sklearn_all_cores)
    import contextlib

    limit = contextlib.nullcontext()
    if allowed:
        try:
            from threadpoolctl import threadpool_limits

            limit = threadpool_limits(limits=cores)         # BLAS / OpenMP pools sized for the allowance, not for the visible CPUs
        except Exception:
            pass
    t0 = time.perf_counter()
    with warnings.catch_warnings(), limit:
        warnings.simplefilter("ignore")
        o = orc.OracleClassifier(**okw).fit(sample)
    dt = time.perf_counter() - t0
    per_iter = (dt - o.timings["prologue"]) / iters
    full_fit = o.timings["prologue"] + per_iter * args.iters     # iterations are identical work
    return {"value": round(n / full_fit, 2), "unit": "cells/s", "cores": cores, "cpus_visible": os.cpu_count(), "cpu_allowance": allowed, "kind": "port",
            "sample": f"{n} of {X.shape[0]} cells x {X.shape[1]} genes, {iters} iterations timed ({dt:.1f} s) and scaled to "
                      f"n_iters={args.iters}; dense log matrix + sklearn randomized PCA + exact kNN + host Louvain.  "
                      + ("All cells (SURVEY.md section 8 d)." if args.cpu_full else
                         "A row sample flatters the CPU: kNN and the Jaccard graph grow faster than linearly in the number of "
                         "cells, so cells/s at the full size is lower than this figure (profiles/r03_cpu_full.json: the full-size run)"),
            "stage_seconds": {k2: round(v, 3) for k2, v in o.timings.items()}}


if __name__ == "__main__":
    _a = parse()
    if _a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(_a.gpus))
    main()
