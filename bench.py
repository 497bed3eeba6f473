#!/usr/bin/env python3
"""bench.py -- exact top-10 QPS of the vfsx single-query scan on MI355X.

Contract (see the task's bench rules): `python bench.py --gpus N --steps K --warmup W`.
  * N = 1: workload C2 = BASELINE.json configs[1]: 1M x 768 f32 Cosine, exact top-10,
    single-query scans.  A step = `--queries-per-step` distinct queries, each scanned
    by its own kernel launch (the corpus is re-read from HBM for every query).
  * N > 1: workload C5 = BASELINE.json configs[4] cut to N ranks: every rank holds a
    12.5M x 768 f32 shard (38.4 GB) of a row-sharded N*12.5M-row corpus -- at N = 8 the
    100M x 768 Euclidean configuration.  Every query is scanned on every shard, the per-shard
    top-10 lists of a step are exchanged with ONE RCCL all-gather (the library's own
    communicator, vfsx_comm_*: ncclAllGather + device merge by (distance, global row)).
    `value` = queries per second answered against the WHOLE corpus.  Weak scaling: the corpus
    grows with N, so ideal scaling is a constant `value` (and N x the aggregate HBM rate, which
    `roofline` reports against N x 8 TB/s).  `--workload c2` selects 1M-row shards instead.
    Started from a plain shell (`python bench.py --gpus N`, WORLD_SIZE unset) the script
    launches its own N ranks through torch.distributed.run and relays rank 0's JSON line; under
    torch.distributed.run it is one of the ranks.
Inputs are resident in HBM before the timed region (synthetic, generated on device).
"""
from __future__ import annotations

import argparse
import datetime
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16
MFMA_PEAK_I8_TOPS = 5000.0  # same guide: the I8 forms run at the cycles of the BF16 form at twice the K

WORKLOADS = {
    # name: (rows per GPU, dim, quantize, metric, queries per step, label)
    "c2": (1_000_000, 768, False, 1, 256, "C2 1Mx768 f32 Cosine exact top-10, single-query scan"),
    "c2e": (1_000_000, 768, False, 0, 256, "1Mx768 f32 Euclidean exact top-10, single-query scan"),
    "c3": (10_000_000, 128, True, 0, 512, "C3 10Mx128 int8 QuantizedVector Euclidean exact top-10"),
    "c3i": (10_000_000, 128, True, 4, 512, "10Mx128 int8, LABELLED integer-domain fast mode (extension, not the "
                                            "reference's arithmetic; SURVEY 8a A6(ii)), exact top-10"),
    "c1": (100_000, 128, False, 2, 4096, "C1 100kx128 f32 SimdEuclidean exact top-10"),
    "c1d64": (100_000, 64, False, 2, 4096, "100kx64 f32 SimdEuclidean exact top-10 (the widest dimension the reference's "
                                           "own Simd path accepts, rank.rs:303)"),
    "c5": (12_500_000, 768, False, 0, 32, "C5 shard 12.5Mx768 f32 Euclidean per GPU (100Mx768 over 8 GPUs), "
                                          "single-query scan, RCCL all-gather top-k merge"),
    "c4": (10_000_000, 1024, False, 1, 4096, "C4 4096-query batch x 10Mx1024 f32 Cosine as Q.X^T on bf16 MFMA (bf16 mirror, "
                                             "the operand BASELINE config 4 names) + fused top-k, exact f32 re-score"),
    "c4e": (10_000_000, 1024, False, 0, 4096, "4096-query batch x 10Mx1024 f32 Euclidean as Q.X^T on bf16 MFMA + per-row "
                                              "upper-bound epilogue + fused top-k, exact f32 re-score (extension of C4 to "
                                              "the Euclidean metric)"),
    "c4auto": (10_000_000, 1024, False, 1, 4096, "C4 with the operand the library picks by itself for an f32 store: an int8 "
                                                 "mirror with per-row scales on int8 MFMA unless the data says bf16; same "
                                                 "exact results"),
    "c4eauto": (10_000_000, 1024, False, 0, 4096, "C4 Euclidean with the library's own choice of operand (int8 mirror)"),
    "c4i8": (10_000_000, 1024, True, 1, 4096, "4096-query batch x 10Mx1024 int8 QuantizedVector Cosine: int8 MFMA over the "
                                              "rb64 corpus itself (no mirror) + fused top-k, exact f32 re-score"),
    "c2m": (1_000_000, 768, False, 1, 256, "C2 1Mx768 f32 Cosine exact top-10, ONE query per call, with the opt-in int8 "
                                           "mirror (+1 B/element): int8 MFMA prefilter over the mirror, exact f32 "
                                           "re-score + safety check, exact scan for flagged queries; same results as c2"),
    "c5m": (12_500_000, 768, False, 0, 32, "C5 shard 12.5Mx768 f32 Euclidean, ONE query per call, with the opt-in int8 "
                                           "mirror (+1 B/element); same results as c5"),
    "c3b": (10_000_000, 128, True, 0, 4096, "C3 in batches: 4096 queries x 10Mx128 int8 QuantizedVector Euclidean, int8 "
                                            "MFMA over the rb64 corpus + upper-bound epilogue, exact f32 re-score"),
}
MFMA_WORKLOADS = ("c4", "c4e", "c4auto", "c4eauto", "c4i8", "c3b")
# single queries through the mirror: on an f32 store the prefilter reads 1 B per element where the exact scan reads 4
MIRROR_SINGLE_WORKLOADS = ("c2m", "c5m")
METRIC_NAMES = ["euclidean", "cosine", "simd_euclidean", "simd_cosine", "euclidean_i8_domain"]


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: c2 on one GPU, c5 (12.5M-row shard per rank) on several")
    ap.add_argument("--queries-per-step", type=int, default=0, help="default: per workload, ~0.1 s of device time")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--mode", default="single", choices=["single", "batch"],
                    help="single: one scan launch per query (the headline); batch: one call per step, "
                         "queries share corpus passes (8 per pass)")
    ap.add_argument("--rows", type=int, default=0, help="override rows per GPU")
    ap.add_argument("--dim", type=int, default=0, help="override the dimension (experiments only)")
    ap.add_argument("--metric", type=int, default=-1, help="override the metric 0..3 (experiments only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=262144)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline: stop after about this long")
    ap.add_argument("--dry-run", action="store_true",
                    help="join the ranks, exchange one collective, print the configuration; touches no GPU "
                         "(launcher rehearsal on CPU with VFSX_BENCH_BACKEND=gloo)")
    return ap.parse_args(argv)


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n: int, argv) -> int:
    """`python bench.py --gpus N` from a plain shell: start the N ranks as children of this process (which has
    touched no GPU and does not import torch) and relay their output -- rank 0 prints the JSON line -- and exit
    code.  The driver's own form (torch.distributed.run ... bench.py) lands in main() directly."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args.gpus, argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    workload = args.workload or ("c2" if world == 1 else "c5")
    rows, dim, quantize, metric, qps_default, label = WORKLOADS[workload]
    mfma = workload in MFMA_WORKLOADS
    mirror_single = workload in MIRROR_SINGLE_WORKLOADS
    if mfma:
        args.mode = "batch"
        if args.steps == 40:  # ~2 s of timed region at 40-70 ms per 4096-query batch
            args.steps, args.warmup = 30, 2
    if args.rows:
        rows = args.rows
    if args.dim:
        dim, label = args.dim, label + f" [dim overridden to {args.dim}]"
    if args.metric >= 0:
        metric, label = args.metric, label + f" [metric overridden to {args.metric}]"
    k = args.k
    qps_step = args.queries_per_step or qps_default
    elem = 1 if quantize else 4
    config = {"workload": label, "rows_per_gpu": rows, "rows_total": rows * world, "dim": dim, "k": k,
              "storage": "i8 (QuantizedVector, dequantized on read)" if quantize else "f32",
              "queries_per_step": qps_step, "mode": args.mode, "metric_fn": METRIC_NAMES[metric],
              "sharding": f"row-sharded x{world}, all-gather top-k merge" if world > 1 else "single GPU"}

    import torch
    import torch.distributed as dist

    backend = os.environ.get("VFSX_BENCH_BACKEND", "nccl")  # nccl == RCCL on ROCm; gloo for rehearsals
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

    # every wait on a peer is bounded: a rank that never arrives fails the job instead of hanging it
    pg_timeout = datetime.timedelta(seconds=int(os.environ.get("VFSX_BENCH_PG_TIMEOUT_S", "600")))
    fault = os.environ.get("VFSX_BENCH_FAULT", "")  # rehearsals: "exit_before_agree:<rank>"

    if args.dry_run:
        joined, preflight = 1, None
        if world > 1:
            use_nccl = backend == "nccl" and torch.cuda.is_available()
            dist.init_process_group("nccl" if use_nccl else "gloo", timeout=pg_timeout)
            if fault == f"exit_before_agree:{rank}":
                os._exit(7)  # a rank that dies before the communicator is agreed on
            t = torch.ones(1, dtype=torch.int64)
            if use_nccl:
                t = t.cuda(local_rank % torch.cuda.device_count())
            dist.all_reduce(t)
            joined = int(t.item())
            # the agreement that precedes vfsx_comm_init in a real run (pre-flight of every rank + rank 0's id); no
            # rank enters ncclCommInitRank here.  Without GPUs every rank reports "device not visible".
            from vfs_amd import sharded
            try:
                sharded.Comm.agree(local_rank)
                preflight = "ok"
            except RuntimeError as e:
                preflight = str(e)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "ranks_joined": joined, "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "comm_preflight": preflight, "config": config}), flush=True)
        return 0

    if not torch.cuda.is_available():
        print("bench.py needs a GPU (vfs_amd has no CPU path)", file=sys.stderr)
        return 2
    import vfs_amd
    from vfs_amd import sharded, synth

    # one rank per GPU; the modulo only matters when rehearsing N ranks on fewer GPUs
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm = None
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout)
            if fault == f"exit_before_agree:{rank}":
                os._exit(7)
            # first contact: one tiny all-reduce on torch's communicator (a rank that cannot reach its peers fails
            # HERE, within pg_timeout), then the library's own RCCL communicator (vfsx_comm_*).  Comm.bootstrap
            # agrees over the group that every rank passed its pre-flight BEFORE anyone enters ncclCommInitRank,
            # so its exception is raised on all ranks or on none; vfsx_comm_init bounds the rest by its watchdog.
            t = torch.ones(1, dtype=torch.int32, device=dev)
            dist.all_reduce(t)
            if int(t.item()) != world:
                print(f"[bench rank {rank}] first all-reduce returned {int(t.item())}, expected {world}", file=sys.stderr)
                return 3
            try:
                comm = sharded.Comm.bootstrap(local_rank)
                comm.reserve(qps_step * k)  # the exchange never allocates inside the timed region
            except Exception as e:  # noqa: BLE001
                print(f"[bench rank {rank}] vfsx_comm bootstrap failed ({e}); torch.distributed all-gather instead",
                      file=sys.stderr, flush=True)
                comm = None
            ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # every rank takes the same path
            if int(ok.item()) == 0 and comm is not None:
                comm.close()
                comm = None
        else:
            dist.init_process_group(backend, timeout=pg_timeout)
            if fault == f"exit_before_agree:{rank}":
                os._exit(7)

    debug = bool(os.environ.get("VFSX_BENCH_DEBUG"))

    def dbg(msg):
        if debug:
            print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    dbg("process group up")
    corpus = vfs_amd.Corpus(dim, quantize=quantize, capacity_rows=rows, device=local_rank)
    corpus.fill_synthetic(synth.CORPUS_SEED, rows, row0=rank * rows, id0=1 + rank * rows)
    corpus.set_row_base(rank * rows)
    if mfma or mirror_single:
        corpus.enable_mfma(True, mirror="bf16" if workload in ("c4", "c4e") else "auto")
        mirror_kind, mirror_rx = corpus.mfma_mirror()

    dbg("corpus filled")
    n_q = (args.steps + args.warmup) * qps_step
    q_host = synth.rows_f32(synth.QUERY_SEED, 0, n_q, dim)
    d_q = torch.from_numpy(q_host).to(dev)
    hit_words = 6  # vfsx_hit = 24 bytes = 6 x u32
    d_local = torch.zeros((qps_step, k, hit_words), dtype=torch.int32, device=dev)
    d_global = torch.zeros((qps_step, k, hit_words), dtype=torch.int32, device=dev)

    def run_step(step_idx: int) -> None:
        stream = torch.cuda.current_stream().cuda_stream
        base = step_idx * qps_step
        if args.mode == "batch":
            corpus.search_device(d_q[base].data_ptr(), qps_step, k, metric, d_local.data_ptr(), stream)
        else:
            for j in range(qps_step):  # one scan launch per query: "single-query scan"
                corpus.search_device(d_q[base + j].data_ptr(), 1, k, metric, d_local[j].data_ptr(), stream)
        if comm is not None:  # ONE ncclAllGather per step (queries_per_step x k x 24 B per rank) + device merge
            comm.merge_hits(d_local.data_ptr(), qps_step, k, d_global.data_ptr(), stream)
        elif world > 1:  # gloo rehearsal
            sharded.exchange_and_merge(d_local, k, device_index=local_rank)

    def fence() -> None:
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for s in range(args.warmup):
        run_step(s)
        dbg(f"warmup step {s} enqueued")
    fence()
    dbg("warmup done")
    t0 = time.perf_counter()
    for s in range(args.steps):
        run_step(args.warmup + s)
    fence()
    elapsed = time.perf_counter() - t0
    dbg("timed region done")
    reduce_dev = dev if world > 1 and dist.get_backend() == "nccl" else "cpu"
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    dbg("max over ranks done")

    # N > 1: the same steps without the exchange (every rank scans its shard, nobody merges): the rate one shard
    # alone sustains in THIS run.  value / that = what the collective + merge + rank skew cost, i.e. the weak-scaling
    # efficiency of the query path (ideal: value stays at the one-shard rate while the corpus grows N-fold).
    local_only_qps = None
    if world > 1:
        def local_step(step_idx: int) -> None:
            stream = torch.cuda.current_stream().cuda_stream
            base = step_idx * qps_step
            if args.mode == "batch":
                corpus.search_device(d_q[base].data_ptr(), qps_step, k, metric, d_local.data_ptr(), stream)
            else:
                for j in range(qps_step):
                    corpus.search_device(d_q[base + j].data_ptr(), 1, k, metric, d_local[j].data_ptr(), stream)
        fence()
        t1 = time.perf_counter()
        for s_i in range(args.steps):
            local_step(args.warmup + s_i)
        fence()
        t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=reduce_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        local_only_qps = args.steps * qps_step / float(t.item())

    # roofline leg: same launches, scan kernel bracketed by HIP events on its own stream
    roof_steps = min(args.steps, 4)
    corpus.set_profiling(True)
    for s in range(roof_steps):
        base = (args.warmup + s) * qps_step
        stream = torch.cuda.current_stream().cuda_stream
        if args.mode == "batch":
            corpus.search_device(d_q[base].data_ptr(), qps_step, k, metric, d_local.data_ptr(), stream)
        else:
            for j in range(qps_step):
                corpus.search_device(d_q[base + j].data_ptr(), 1, k, metric, d_local[j].data_ptr(), stream)
    torch.cuda.synchronize()
    scan_ms, scan_launches = corpus.scan_time_ms()
    corpus.set_profiling(False)
    dbg("roofline leg done")
    avg_ms = scan_ms / max(scan_launches, 1)
    if world > 1:  # the slowest rank's kernel bounds the job
        t = torch.tensor([avg_ms], dtype=torch.float64, device=reduce_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        avg_ms = float(t.item())
    algo_bytes = rows * dim * elem  # N*d*sizeof(elem) per query and GPU, ids/norms excluded
    achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
    if mfma:  # flops = 2 * nq * N * d per batch; the dominant kernel is the prefilter GEMM
        launches_per_batch = max(1, round(scan_launches / max(roof_steps, 1)))
        flops = 2.0 * qps_step * rows * dim / launches_per_batch
        tflops = flops / (avg_ms * 1e-3) / 1e12
        mq, mf = corpus.mfma_stats()

    # per-query latency (>=100 back-to-back distinct queries, median), single mode only
    latency = None
    if args.mode == "single" and world == 1:
        n_lat = min(128, n_q)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_lat + 1)]
        stream = torch.cuda.current_stream().cuda_stream
        evs[0].record()
        for j in range(n_lat):
            corpus.search_device(d_q[j].data_ptr(), 1, k, metric, d_local[0].data_ptr(), stream)
            evs[j + 1].record()
        torch.cuda.synchronize()
        lat = sorted(evs[j].elapsed_time(evs[j + 1]) for j in range(n_lat))
        latency = {"queries": n_lat, "median": round(lat[n_lat // 2], 4), "p99": round(lat[min(n_lat - 1, int(n_lat * 0.99))], 4),
                   "min": round(lat[0], 4), "unit": "ms", "what": "device time per query (scan + select), back-to-back"}

    queries = args.steps * qps_step
    qps_global = queries / elapsed
    # which scan kernel vfsx_search_device picks (vfsx_api.cpp exact_search_device): the 16-wave int8 kernel for
    # single-query scans of int8 stores with >= 16 row-blocks per CU and k <= 64
    scan_kernel_name = "scan_kernel"
    if metric == 4:
        scan_kernel_name = "scan_i8dom_kernel"
    elif (quantize and args.mode == "single" and k <= 64 and os.environ.get("VFSX_I8_WIDE", "1") != "0"
          and (rows + 63) // 64 >= 16 * torch.cuda.get_device_properties(local_rank).multi_processor_count):
        scan_kernel_name = "scan_i8w_kernel"
    traffic, traffic_source = pmc_traffic(workload, rows)
    out = {
        "metric": (f"exact top-{k} QPS (query batches on the matrix cores + exact re-score; queries/s)" if mfma else
                   f"exact top-{k} QPS (single-query scan; queries/s against the whole row-sharded corpus)"),
        "value": round(qps_global, 3),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": config,
        "timed_region_s": round(elapsed, 3),
        "shard_scans_per_s": round(qps_global * world, 3),
        "hbm_gbps_aggregate": round(qps_global * algo_bytes * world / 1e9, 1),
        "hbm_frac_of_aggregate_peak": round(qps_global * algo_bytes / 1e9 / HBM_PEAK_GBPS, 4),
        "roofline": {"bound": "hbm", "achieved": round(achieved * world, 1), "peak": HBM_PEAK_GBPS * world, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": scan_kernel_name, "avg_kernel_ms": round(avg_ms, 5),
                     "algorithmic_bytes_per_launch": algo_bytes, "launches_timed": int(scan_launches),
                     "what": ("slowest rank's average scan-kernel launch (HIP events on its stream); achieved / peak are "
                              "the aggregate over the ranks") if world > 1 else
                             "average scan-kernel launch (HIP events on its stream)"},
    }
    if world > 1:
        out["scaling_note"] = ("rows_per_gpu is fixed, so the corpus grows with N: ideal weak scaling is a constant "
                               "value (queries/s against the whole corpus) and N x hbm_gbps_aggregate")
        out["one_shard_rate_in_this_run"] = {"value": round(local_only_qps, 3), "unit": "queries/s",
                                             "what": "the same steps with no exchange (slowest rank): the ideal for value"}
        out["weak_scaling_efficiency_vs_one_shard"] = round(qps_global / local_only_qps, 4)
        out["rccl_ranks"] = (int(vfs_amd.load().vfsx_comm_world(comm.handle)) if comm is not None else
                             (dist.get_world_size() if dist.get_backend() == "nccl" else None))
        out["collective"] = ("vfsx_comm_merge_hits: ncclAllGather (RCCL) + device merge, one per step" if comm is not None
                             else ("torch.distributed all_gather_into_tensor (RCCL) + device merge, one per step "
                                   "(vfsx_comm bootstrap failed: see stderr)" if dist.get_backend() == "nccl" else
                                   f"torch.distributed {dist.get_backend()} rehearsal (not RCCL)"))
    if metric == 4:
        out["dtype"] = "int32 dot4 on i8 (integer-domain extension) + f32 sqrt/div"
    if latency:
        out["query_latency_ms"] = latency
    props = torch.cuda.get_device_properties(local_rank)
    out["device"] = {"name": props.name, "arch": getattr(props, "gcnArchName", ""), "cus": props.multi_processor_count,
                     "hbm_gib": round(props.total_memory / 2**30, 1),
                     "peak_source": "MI355X_MICROARCH.md (8 TB/s HBM3E, 2.5 PFLOP/s dense bf16)"}
    if mfma:
        i8_mfma = mirror_kind in ("int8", "int8-store")
        mfma_peak = MFMA_PEAK_I8_TOPS if i8_mfma else MFMA_PEAK_TFLOPS
        out["roofline"] = {"bound": "mfma", "achieved": round(tflops, 1), "peak": mfma_peak,
                           "unit": "TOP/s" if i8_mfma else "TFLOP/s",
                           "frac": round(tflops / mfma_peak, 4), "traffic": traffic, "traffic_source": traffic_source,
                           "kernel": "mfma_prefilter256_kernel", "avg_kernel_ms": round(avg_ms, 4),
                           "algorithmic_flops_per_launch": flops, "launches_timed": int(scan_launches)}
        for key in ("hbm_gbps_aggregate", "hbm_frac_of_aggregate_peak", "shard_scans_per_s"):
            out.pop(key, None)  # HBM figures describe the single-query scan, not a matrix-core batch
        out["mfma_queries"] = int(mq)
        out["mfma_fallback_queries"] = int(mf)
        out["dtype"] = ("int8 MFMA (i32 accumulate) prefilter + f32 exact re-score" if i8_mfma else
                        "bf16 MFMA prefilter + f32 exact re-score")
        out["mfma_operand"] = {"kind": mirror_kind, "measured_row_residual": round(mirror_rx, 6),
                               "after_run": corpus.mfma_mirror()[0]}

    if mirror_single:
        # the profiled kernel is the small-batch prefilter: HBM-bound on the MIRROR's bytes (1 B per element, int8;
        # 2 B, bf16), not on the store's 4 B per element
        mq, mf = corpus.mfma_stats()
        m_elem = 1 if mirror_kind == "int8" else 2
        m_bytes = rows * dim * m_elem
        m_ach = m_bytes / (avg_ms * 1e-3) / 1e9
        out["metric"] = (f"exact top-{k} QPS (one query per call; int8-mirror prefilter on the matrix cores + exact f32 "
                         "re-score; queries/s against the whole row-sharded corpus)")
        out["dtype"] = "int8 MFMA (i32 accumulate) prefilter + f32 exact re-score" if m_elem == 1 else \
                       "bf16 MFMA prefilter + f32 exact re-score"
        out["roofline"] = {"bound": "hbm", "achieved": round(m_ach * world, 1), "peak": HBM_PEAK_GBPS * world,
                           "unit": "GB/s", "frac": round(m_ach / HBM_PEAK_GBPS, 4), "traffic": None,
                           "kernel": "mfma_prefilter_skinny_kernel", "avg_kernel_ms": round(avg_ms, 5),
                           "algorithmic_bytes_per_launch": m_bytes, "launches_timed": int(scan_launches),
                           "what": "average prefilter launch (HIP events on its stream) over the mirror's bytes"}
        for key in ("hbm_gbps_aggregate", "hbm_frac_of_aggregate_peak", "shard_scans_per_s"):
            out.pop(key, None)  # those describe the 4 B per element scan
        out["mfma_queries"] = int(mq)
        out["mfma_fallback_queries"] = int(mf)
        out["mfma_operand"] = {"kind": mirror_kind, "measured_row_residual": round(mirror_rx, 6),
                               "after_run": corpus.mfma_mirror()[0]}
        if latency:
            latency["what"] = "device time per query (prefilter + select + re-score [+ exact scan if flagged]), back-to-back"

    # Side leg (not `value`): the same single-query steps with the opt-in int8 mirror switched on -- the prefilter
    # reads 1 B per element where the scan above reads 4, the hits are the same bits (checked here on one step).
    if (args.mode == "single" and not quantize and not mirror_single and not mfma and rows * dim >= 175_000_000
            and k <= 32 and metric in (0, 1, 2, 3) and not os.environ.get("VFSX_BENCH_NO_MIRROR_LEG")
            and (world == 1 or os.environ.get("VFSX_BENCH_MIRROR_LEG"))):  # N > 1: only on request (rehearsals)
        try:
            check_step = args.warmup
            run_step(check_step)
            fence()
            ref_hits = (d_global if world > 1 else d_local).clone()
            corpus.enable_mfma(True)
            corpus.mfma_stats(reset=True)
            run_step(check_step)  # builds the mirror; same queries as the reference step
            fence()
            same = bool(torch.equal(ref_hits, d_global if world > 1 else d_local))
            m_steps = max(1, min(args.steps, 10))
            tm = time.perf_counter()
            for s_i in range(m_steps):
                run_step(args.warmup + s_i)
            fence()
            tm = time.perf_counter() - tm
            if world > 1:
                t = torch.tensor([tm], dtype=torch.float64, device=reduce_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                tm = float(t.item())
            mq, mf = corpus.mfma_stats()
            out["with_int8_mirror"] = {
                "value": round(m_steps * qps_step / tm, 3), "unit": "queries/s", "steps": m_steps,
                "hits_identical_to_exact_scan": same, "mfma_queries": int(mq), "mfma_fallback_queries": int(mf),
                "mirror": corpus.mfma_mirror()[0],
                "what": "same steps, one query per call, vfsx_corpus_enable_mfma on (+1 B/element): int8 MFMA prefilter "
                        "over the mirror + exact f32 re-score + safety check (bench.py --workload c2m / c5m)"}
        except Exception as e:  # noqa: BLE001 -- a side leg must not cost the line
            if world > 1:
                raise
            out["with_int8_mirror"] = {"error": f"{type(e).__name__}: {e}"}
        corpus.enable_mfma(False)

    # Side leg (not `value`): what ONE GPU does on the shard the N > 1 runs give every rank (12.5M x 768 f32 Euclidean,
    # 38.4 GB).  The driver's N = 1 line is C2 and its N > 1 lines are C5 shards, so this is the number the N > 1
    # `value`s are to be read against: ideal weak scaling is a constant `value` equal to it.
    if (world == 1 and workload == "c2" and not args.rows and not args.dim and args.metric < 0
            and not os.environ.get("VFSX_BENCH_NO_C5_LEG")):
        try:
            out["c5_shard_one_gpu"] = c5_shard_leg(vfs_amd, synth, torch, dev, local_rank, k)
        except Exception as e:  # noqa: BLE001 -- a side leg must not cost the line
            out["c5_shard_one_gpu"] = {"error": f"{type(e).__name__}: {e}"}
        out["scaling_note"] = ("this N = 1 line is BASELINE config 2 (1M rows); `--gpus N` lines (N > 1) are weak scaling "
                               "on BASELINE config 5's shard, 12.5M x 768 f32 Euclidean per GPU: their ideal `value` is "
                               "c5_shard_one_gpu.value (the corpus grows with N, the query rate stays), NOT this line's value")

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, rows, dim, quantize, metric, k)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    corpus.close()
    dbg("closed")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def c5_shard_leg(vfs_amd, synth, torch, dev, device_index, k):
    """One C5 shard (WORKLOADS["c5"]) on this GPU: 4 untimed + 128 timed single-query Euclidean scans, wall-clocked
    like the main region, plus the scan kernel's average from HIP events on its stream."""
    rows, dim, _, metric, _, label = WORKLOADS["c5"]
    n_q, n_warm = 128, 4
    shard = vfs_amd.Corpus(dim, capacity_rows=rows, device=device_index)
    try:
        shard.fill_synthetic(synth.CORPUS_SEED, rows, row0=0, id0=1)
        d_q = torch.from_numpy(synth.rows_f32(synth.QUERY_SEED, 0, n_q + n_warm, dim)).to(dev)
        d_hits = torch.zeros((n_q + n_warm, k, 6), dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream().cuda_stream

        def scans(lo, hi):
            for j in range(lo, hi):
                shard.search_device(d_q[j].data_ptr(), 1, k, metric, d_hits[j].data_ptr(), stream)
            torch.cuda.synchronize()

        scans(0, n_warm)
        t0 = time.perf_counter()
        scans(n_warm, n_warm + n_q)
        dt = time.perf_counter() - t0
        shard.set_profiling(True)
        scans(n_warm, n_warm + 16)
        ms, launches = shard.scan_time_ms()
        shard.set_profiling(False)
        avg_ms = ms / max(launches, 1)
        gbps = rows * dim * 4 / (avg_ms * 1e-3) / 1e9
        return {"value": round(n_q / dt, 3), "unit": "queries/s", "queries": n_q, "avg_kernel_ms": round(avg_ms, 5),
                "achieved_gbps": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBPS, 4), "workload": label,
                "what": "measured in this run after the timed region: one 12.5M x 768 f32 shard, one scan launch per "
                        "query; the ideal `value` of every `--gpus N` line (N x this many shard scans per second)"}
    finally:
        shard.close()


def rust_nightly_leg():
    """SURVEY 8(d) / BASELINE.md: the reference's own nightly core::simd path can only be timed where a Rust
    nightly toolchain exists.  Probe it; say so when it does not."""
    try:
        r = subprocess.run(["cargo", "+nightly", "--version"], capture_output=True, text=True, timeout=20)
    except FileNotFoundError:
        return "unavailable (no cargo on this box: the reference's nightly core::simd path cannot be built here)"
    except (OSError, subprocess.SubprocessError) as e:
        return f"unavailable (cargo +nightly --version: {type(e).__name__})"
    if r.returncode != 0:
        return "unavailable (cargo present, no nightly toolchain: " + (r.stderr.strip().splitlines() or ["?"])[-1][:120] + ")"
    crate = os.path.join(ROOT, "rust", "simd_baseline")
    try:
        b = subprocess.run(["cargo", "+nightly", "run", "--release", "--offline", "--quiet"], cwd=crate,
                           capture_output=True, text=True, timeout=300)
    except (OSError, subprocess.SubprocessError) as e:
        return f"unavailable (rust/simd_baseline did not run: {type(e).__name__})"
    if b.returncode != 0:
        return "unavailable (rust/simd_baseline did not build: " + (b.stderr.strip().splitlines() or ["?"])[-1][:160] + ")"
    try:
        return json.loads(b.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return "unavailable (rust/simd_baseline printed no JSON line)"


def pmc_traffic(workload, rows):
    """HBM bytes per dominant-kernel launch from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN/traffic.json, written by tools/summarize_profile.py) and the file it came from; (None, None) if
    there is no profile for this workload/size.  PMC counters cannot be read in-process: this is a REPLAYED
    figure, labelled as such by `traffic_source`."""
    import glob
    if rows != WORKLOADS[workload][0]:
        return None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            if workload in t:
                return (t[workload]["scan_hbm_bytes_per_launch"],
                        os.path.relpath(path, ROOT) + " (replayed from the committed PMC passes, not measured in this run)")
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def cpu_baseline(args, rows, dim, quantize, metric, k):
    """The oracle (oracle/vfs_oracle.c: the reference's scalar arithmetic, sequential
    f32 sums, one core) timed on a bounded sample of the same workload."""
    from oracle import c_oracle
    from vfs_amd import synth
    n = min(args.cpu_sample_rows, rows)
    x = synth.rows_f32(synth.CORPUS_SEED, 0, n, dim)
    if quantize:
        x = c_oracle.quantize(x, 127.0)
    qs = synth.rows_f32(synth.QUERY_SEED, 0, 4096, dim)
    c_oracle.topk_scan(x[:1024], qs[0], metric, k)  # warm
    t0 = time.perf_counter()
    done = 0
    for q in qs:
        c_oracle.topk_scan(x, q, metric, k)
        done += 1
        if time.perf_counter() - t0 >= args.cpu_seconds:
            break
    dt = time.perf_counter() - t0
    rows_per_s = n * done / dt
    out = {"value": round(rows_per_s / rows, 4), "unit": "queries/s", "cores": 1, "kind": "port",
           "sample": f"{done} queries x first {n} rows of the same synthetic corpus "
                     f"({dt:.1f} s of CPU), rows/s scaled to the full {rows}-row corpus",
           "rows_per_s": round(rows_per_s, 1)}
    out["rust_nightly"] = rust_nightly_leg()
    if metric in (2, 3) and dim not in (2, 4, 8, 16, 32, 64):
        out["note"] = ("the reference's Simd* methods panic unless d is one of 2..64 (rank.rs:31,303,307); this is the "
                       "scalar-order port, which Simd* equals bit for bit wherever it runs (ordered reduce_sum)")
    # context only: a 64-lane-chunked, vectorised restatement of the
    # reference's core::simd path (oracle/cpu_context.c; different summation order, so NOT the
    # parity arithmetic), on one core and on this process's cores (capped at 16)
    if metric > 3:
        return out
    try:
        threads = min(16, len(os.sched_getaffinity(0)))
    except AttributeError:
        threads = min(16, os.cpu_count() or 1)
    ctx = {"arithmetic": "64-lane chunked sums (not the reference's order); timing context only",
           "avx2": bool(c_oracle.lib().vfsc_have_avx2())}
    for label, nt in (("simd_chunked_1core", 1), ("simd_chunked_allcores", threads)):
        c_oracle.chunked_scan(x[:1024], qs[0], metric, k, nt)
        t0 = time.perf_counter()
        nq = 0
        for q in qs:
            c_oracle.chunked_scan(x, q, metric, k, nt)
            nq += 1
            if time.perf_counter() - t0 >= args.cpu_seconds / 3:
                break
        dtc = time.perf_counter() - t0
        ctx[label] = {"value": round(n * nq / dtc / rows, 4), "unit": "queries/s", "cores": nt}
    out["context"] = ctx
    return out


if __name__ == "__main__":
    sys.exit(main())
