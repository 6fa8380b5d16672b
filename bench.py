#!/usr/bin/env python3
"""bench.py -- headline benchmark of the trace loop: book-1 final scene, 1200x800, 500 spp, depth 50 (BASELINE configuration 2).

One step = one full render of that frame.  Inputs (sphere table, camera) are resident in HBM before the timed region starts.

  N = 1   `python bench.py`: one GPU traces all 500 samples of every pixel and resolves to 8 bits.
  N > 1   STRONG scaling of the same frame (BASELINE's headline: 500 spp on N GPUs): the samples of every pixel are split over the
          GPUs by global sample index, ONE RCCL all-reduce sums the float accumulation buffers over xGMI, rank / member 0
          resolves.  Two hosts run it, chosen before the first GPU call:
            * `python bench.py --gpus N` (no launcher, WORLD_SIZE unset): ONE process drives N devices through the C ABI's
              rtx_group_* calls (ncclCommInitAll + ncclAllReduce inside librtx.so) -- the path renderSceneHIP / rtx_cli -G use;
            * `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
              --gpus N ...` (WORLD_SIZE = N): one process per GPU, torch.distributed backend "nccl" (= RCCL).
          The weak-scaling figure (500 spp PER GPU, a 500 N-spp image), the exchange's own time and the per-GPU trace-kernel
          times ride along as extra keys; --weak makes the weak figure the headline instead.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--spp S] [--weak] [--no-cpu-baseline] [--no-extras]

Rank 0 prints ONE JSON line (metric Msamples/s; Mrays/s, the roofline of the trace kernel -- f16 MFMA flop issued against
the dense f16 peak, plus the counter-derived vector-issue utilisation -- and the CPU baseline ride in the same object).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

COLS, ROWS, DEPTH, SEED = 1200, 800, 50, 1
FP32_PEAK_TFLOPS = 157.3          # MI355X vector FP32 (MI355X_MICROARCH.md): 256 CU x 2.4 GHz x 256 flop/clk
FLOP_PER_TEST = 18                # sphere.cpp:72-77 with a = d.d hoisted
F16_MFMA_PEAK_TFLOPS = 2500.0     # dense f16 MFMA (MI355X_MICROARCH.md)
MFMA_FLOP_PER_TEST = 64           # filter: 4 x v_mfma_f32_32x32x16_f16 (4 x 32768 flop) per 32 spheres x 64 rays


def host_info():
    """Core count of the machine, the share of it this job may use, and the CPU model.  On the GPU boxes
    os.cpu_count() is the whole machine; the job's share is what OMP_NUM_THREADS (set by the harness) or the affinity mask says."""
    model = "?"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "?")
    except OSError:
        pass
    cores = os.cpu_count() or 1
    share = cores
    try:
        share = min(share, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        share = min(share, max(1, int(os.environ.get("OMP_NUM_THREADS", share))))
    except ValueError:
        pass
    return cores, share, model


def port_vs_reference():
    """The band between this baseline (the port) and the compiled reference's own renderScene at equal thread count, measured
    in the build container by oracle/bench_ref_vs_port.py (the reference cannot travel to the GPU box): a committed constant."""
    try:
        with open(os.path.join(ROOT, "oracle", "ref_vs_port.json")) as f:
            d = json.load(f)
        return {"measured_on": d["host"], "provenance": "oracle/bench_ref_vs_port.py in the build container; " + d["reference"],
                "runs": [{k: r[k] for k in ("threads", "spp", "reference_msamples_per_s", "port_msamples_per_s", "port_over_reference", "psnr_db_8bit_frames")} for r in d["runs"]]}
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(table):
    """The oracle (a port of the reference CPU path: 40-byte AoS spheres, 64x64 tile jobs, iterative colour) timed on this
    box's host cores on BASELINE configuration 1 as written -- 1200x800, 10 spp, depth 50, `-t <all cores of the job's share> -b 64`
    -- and on one thread (the full view at 1200x40, 10 spp)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    host_cores, share, model = host_info()
    cam = pyoracle.book_camera(COLS, ROWS)
    spp, reps, samples, rays, dt = 10, 0, 0, 0, 0.0
    while reps < 6 and (reps < 2 or dt < 8.0):        # configuration 1 is about 2 s on 16 threads: repeated for a stable figure
        t0 = time.perf_counter()
        _, _, st = pyoracle.render(table, cam, ROWS, COLS, 0, spp, DEPTH, SEED, nthreads=share, tile=64)
        dt += time.perf_counter() - t0
        samples += st["samples"]; rays += st["rays"]; reps += 1
    t1 = time.perf_counter()
    _, _, st1 = pyoracle.render(table, cam, 40, COLS, 0, spp, DEPTH, SEED, nthreads=1, tile=40)
    dt1 = time.perf_counter() - t1
    return {"value": round(samples / dt / 1e6, 4), "unit": "Msamples/s", "cores": share, "kind": "port",
            "host_cores": host_cores, "cpu_model": model, "threads_used": share,
            "single_thread": {"value": round(st1["samples"] / dt1 / 1e6, 4), "sample": "the full view at 1200x40, %d spp, 1 thread (%.1f s)" % (spp, dt1)},
            "mrays_per_s": round(rays / dt / 1e6, 4),
            "sample": "BASELINE configuration 1: book1 %dx%d, %d spp, depth %d, 64x64 tile jobs on %d threads, %d repetitions (%.1f s)" % (COLS, ROWS, spp, DEPTH, share, reps, dt),
            "port_vs_reference": port_vs_reference()}


def measured_valu_issue():
    """Vector-issue utilisation of the trace kernel from the committed PMC passes of this same command
    (tools/valu_issue.sh + tools/valu_mix.sh -> tools/valu_issue.py -> profiles/trace_kernel_valu_issue.json).  None when absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "trace_kernel_valu_issue.json")) as f:
            d = json.load(f)
        return {k: d.get(k) for k in ("valu_issue_util_low", "valu_issue_util_high", "rocprof_VALUBusy", "calibration_VALUBusy_of_pure_valu_loops",
                                      "lane_util", "mfma_busy", "mfma_coexec_share", "lds_bank_conflict_share")}
    except (OSError, ValueError):
        return None


def measured_traffic():
    """HBM bytes per trace-kernel launch from the committed PMC passes of this same command (FETCH_SIZE + WRITE_SIZE,
    corrected as MI355X_MICROARCH.md prescribes; tools/rocpd_summary.py writes the file).  None when absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "trace_kernel_traffic.json")) as f:
            return json.load(f).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def other_workloads(ra, device, book1):
    """Side measurements outside the timed region (one GPU, the last of three launches queued back to back, trace-kernel time from HIP events):
    BASELINE configuration 5 (20 164 spheres, 100 spp) with the linear list and with the uniform grid, the
    bench scene with the grid, and the share of the bench frame that one of 8 GPUs traces under strong scaling (62 of 500 spp).
    Informational: `value` above is configuration 2 with the linear list."""
    out = {}
    for name, table, spp in (("config5_20164_spheres_100spp", ra.random_scene(SEED, 71), 100), ("config2_book1_500spp", book1, 500), ("config2_share_of_one_of_8_gpus_62spp", book1, 62)):
        for fname, kind in (("linear_list_mfma", ra.FILTER_MFMA), ("uniform_grid", ra.FILTER_GRID)):
            ctx = ra.Context(device)
            try:
                ctx.set_filter(kind)
                ctx.upload_scene(table)
                ctx.set_camera(ra.book_camera(COLS, ROWS))
                _, st = ctx.render(COLS, ROWS, 0, spp, DEPTH, SEED, want_stats=True)     # counters; the frame's strip costs for the launches below
                for _ in range(3):                       # queued back to back: the last one starts on a busy GPU, as in a render loop
                    ctx.render_async(COLS, ROWS, 0, spp, DEPTH, SEED)
                ctx.sync()
                ms = ctx.last_trace_ms()
                out.setdefault(name, {})[fname] = {"msamples_per_s": round(st["samples"] / ms / 1e3, 1), "kernel_ms": round(ms, 3),
                                                  "exact_tests_per_ray": round(st["candidates"] / max(1, st["rays"]), 2)}
            finally:
                ctx.close()
    return out


def roofline_block(n_spheres, stats, trace_avg_ms):
    """`stats` and `trace_avg_ms` describe ONE launch of the trace kernel (for N > 1: the average member's)."""
    tests = stats["sphere_tests"]
    achieved = FLOP_PER_TEST * tests / (trace_avg_ms * 1e-3) / 1e12
    # what the matrix cores actually execute: every ray meets the table padded to 128 spheres, 64 f16-MFMA flop per pair
    n_pad_mfma = (n_spheres + 127) // 128 * 128
    mfma_tflops = MFMA_FLOP_PER_TEST * stats["rays"] * n_pad_mfma / (trace_avg_ms * 1e-3) / 1e12
    # What the hardware executes for the sphere loop is the f16 MFMA filter: that is the roofline this line is priced
    # against (achieved = filter flop issued on the matrix cores / measured kernel time; dense f16 peak).  The kernel's
    # actual limit is the vector issue port -- sign capture, exact re-tests, samplers and shading share it with the
    # MFMA issue slots -- whose utilisation comes from PMC counters (valu_issue).  BASELINE's north star asks for a
    # fraction of the FP32 VALU peak; that formula (18 flop x rays x n / t) counts arithmetic the filter makes
    # unnecessary and exceeds the peak, so it is reported as a throughput equivalent, not as a fraction.
    return {"bound": "mfma", "kernel": "trace_kernel", "achieved": round(mfma_tflops, 1), "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(mfma_tflops / F16_MFMA_PEAK_TFLOPS, 4), "traffic": measured_traffic(),
            "peak_kind": "dense f16 MFMA (MI355X_MICROARCH.md); flop counted: 64 per (ray, padded sphere) pair = 4 v_mfma_f32_32x32x16_f16 per 64 rays x 32 spheres",
            "kernel_ms": round(trace_avg_ms, 4), "mfma_flop_per_test": MFMA_FLOP_PER_TEST, "padded_spheres": n_pad_mfma,
            "limiting_resource": "vector issue port (VALU + MFMA issue slots)", "valu_issue": measured_valu_issue(),
            "sphere_tests_per_launch": tests, "exact_tests_per_launch": stats["candidates"],
            "fp32_equivalent_tflops": round(achieved, 3),
            "fp32_equivalent_note": "SURVEY s8(d): 18 flop x (rays x n) / kernel time, against the 157.3 TFLOP/s FP32 vector peak the north star names; not executed arithmetic"}


def run_group(args, result_fd):
    """N > 1 in ONE process: N devices through rtx_group_* (librtx.so's own RCCL communicator)."""
    import numpy as np
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (the only mode the host driver supports)
    import raytracing_amd as ra
    n, cols, rows, spp_total = args.gpus, args.cols, args.rows, args.spp
    loopback = os.environ.get("RTX_GROUP_LOOPBACK", "0") not in ("", "0")
    table = ra.random_scene(SEED)
    g = ra.Group([0] * n if loopback else list(range(n)))
    for r in range(n):
        g.member(r).set_filter(ra.FILTER_MFMA)        # the headline is the reference's linear list (see main())
    g.upload_scene(table)
    g.set_camera(ra.book_camera(cols, rows))
    fb = g.member(0).host_framebuffer(cols * rows)    # pinned host memory mapped into GPU 0: the resolve kernel writes it over the host link

    def timed(spp_in_all, steps, warmup):
        for _ in range(warmup):
            g.render_frame(cols, rows, 0, spp_in_all, DEPTH, SEED, spp_in_all, chunk_spp=args.chunk_spp, out=fb)
        members = [[] for _ in range(n)]
        t0 = time.perf_counter()
        for _ in range(steps):
            g.render_frame(cols, rows, 0, spp_in_all, DEPTH, SEED, spp_in_all, chunk_spp=args.chunk_spp, out=fb)   # returns with every member idle
            for r in range(n):
                members[r].append(g.member(r).last_trace_ms())     # HIP events around each member's trace kernel, on its stream (already complete)
        dt = time.perf_counter() - t0
        return dt / steps, [sum(m) / len(m) for m in members]

    strong_spp, weak_spp = spp_total, spp_total * n
    head_spp, side_spp = (weak_spp, strong_spp) if args.weak else (strong_spp, weak_spp)
    stats = g.render(cols, rows, 0, head_spp, DEPTH, SEED, chunk_spp=args.chunk_spp, want_stats=True)   # in-kernel counters, summed over the members
    step_s, member_ms = timed(head_spp, args.steps, args.warmup)
    side_s, side_member_ms = timed(side_spp, max(1, min(args.steps, 3)), 1)
    # the exchange alone: members synchronised first, HIP events on member 0's stream around the RCCL group call
    exchange = {}
    for how in ("allreduce", "reduce"):
        os.environ["RTX_GROUP_EXCHANGE"] = how
        ms = []
        for _ in range(4):
            g.render(cols, rows, 0, n, DEPTH, SEED, want_stats=False)       # (blocks: every member idle)
            g.allreduce()
            ms.append(g.last_exchange_ms())
        exchange[how + "_ms"] = round(min(ms[1:]), 4)
    os.environ["RTX_GROUP_EXCHANGE"] = "allreduce"
    per_launch = {k: stats[k] / n for k in ("rays", "sphere_tests", "candidates")}
    trace_avg_ms = sum(member_ms) / n
    side = {"value": round(cols * rows * side_spp / side_s / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(side_s * 1e3, 4),
            "workload": "%d spp %s" % (side_spp, "in all (strong)" if args.weak else "= %d per GPU (weak)" % spp_total),
            "member_trace_ms": [round(m, 4) for m in side_member_ms]}
    out = {
        "metric": "Msamples/s", "value": round(stats["samples"] / step_s / 1e6, 3), "unit": "Msamples/s",
        "mrays_per_s": round(stats["rays"] / step_s / 1e6, 3),
        "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 4),
        "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "book-1 final scene (seed 1, %d spheres) %dx%d, %s, depth %d" % (len(table), cols, rows, ("%d spp per GPU" % spp_total) if args.weak else ("%d spp in all" % spp_total), DEPTH),
                   "parallelism": ("REHEARSAL (RTX_GROUP_LOOPBACK): %d members on ONE device, exchange = plain sum; not a measurement" % n) if loopback
                                  else "spp split x%d in one process (rtx_group_*: ncclCommInitAll) + RCCL all-reduce of the float accumulation buffer" % n,
                   "rays_per_sample": round(stats["rays"] / stats["samples"], 4),
                   "candidate_search": "linear list: every ray against every sphere through the f16 MFMA filter (RTX_FILTER_MFMA, the north star's path)",
                   "work_order": "cost-ordered from the second launch of the frame on (rtx_last_launch_ordered)"},
        "rccl_ranks": g.size(), "member_trace_ms": [round(m, 4) for m in member_ms], "exchange": exchange,
        "weak_scaling" if not args.weak else "strong_scaling": side,
        "roofline": roofline_block(len(table), per_launch, trace_avg_ms),
    }
    os.write(result_fd, (json.dumps(out) + "\n").encode())
    g.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=500, help="samples per pixel of the frame (with --weak: per GPU)")
    ap.add_argument("--cols", type=int, default=COLS)
    ap.add_argument("--rows", type=int, default=ROWS)
    ap.add_argument("--chunk-spp", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed side measurements (configuration 5, grid filter)")
    ap.add_argument("--strong", action="store_true", help="(the default for N > 1) --spp is the IMAGE's sample count, split over the GPUs by sample index")
    ap.add_argument("--weak", action="store_true", help="weak scaling as the headline: every GPU traces --spp samples of its own range (an N x spp image)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group and all-reduce even with one rank (rehearsal of the N > 1 path)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")

    # stdout carries exactly ONE line, the JSON: libraries that chat on fd 1 (RCCL prints a version banner there) are
    # sent to stderr for the whole run, and the result goes to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    # Which host drives N > 1 is decided here, before anything touches a GPU: a launcher (torch.distributed.run) has set
    # WORLD_SIZE -> one process per GPU; none has -> this one process drives all N devices through rtx_group_*.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return run_group(args, result_fd)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    import torch.distributed as dist
    import raytracing_amd as ra
    from raytracing_amd.multi_gpu import TorchAccum, spp_range

    use_dist = world > 1 or args.force_dist
    if use_dist:                                                 # before the first HIP call: the runtime reads these at start-up
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")            # only reached unset in the 1-rank --force-dist rehearsal
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (the only mode the host driver supports)
    if torch.cuda.device_count() > 0:
        local_rank %= torch.cuda.device_count()      # a launcher may expose one device per rank
    torch.cuda.set_device(local_rank)
    if use_dist:
        # RTX_DIST_BACKEND=gloo: a rehearsal of the N > 1 code on a box with fewer GPUs than ranks (ranks share devices, the sum
        # goes through the host: the number it prints is not a measurement); the driver's runs use "nccl" = RCCL over xGMI
        backend = os.environ.get("RTX_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    cols, rows = args.cols, args.rows
    table = ra.random_scene(SEED)
    ctx = ra.Context(local_rank)
    # BASELINE's north star and configuration 2 name the linear list ("world hit over the sphere list", "sphere table in LDS"):
    # the headline measures THAT path -- every ray against every sphere through the f16 MFMA filter -- although the library's
    # default (RTX_FILTER_AUTO) would walk the uniform grid for a table of this size (identical image, about 25 % faster:
    # other_workloads reports it)
    ctx.set_filter(ra.FILTER_MFMA)
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(cols, rows))
    accum = TorchAccum(ctx, cols, rows, torch.device("cuda", local_rank))
    weak = args.weak or world == 1
    # the host framebuffer (RayTracing.cpp:123-129; its CUDA mode takes it from the GPU runtime too, :125): pinned memory mapped
    # into the GPU, so that the resolve kernel writes it over the host link
    fb = ctx.host_framebuffer(cols * rows) if rank == 0 else None

    def share(weak_mode):
        """(first global sample, samples of this rank, samples of the image)"""
        if weak_mode:                            # rank r owns global samples [r*spp, (r+1)*spp)
            return rank * args.spp, args.spp, args.spp * world
        b, c = spp_range(rank, world, 0, args.spp)   # the image has --spp samples, rank r traces its share of them
        if c == 0:
            raise SystemExit("bench.py: %d samples cannot feed %d ranks" % (args.spp, world))
        return b, c, args.spp

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    def run(weak_mode, steps, warmup):
        spp_begin, spp, spp_total = share(weak_mode)

        def step():
            ctx.render_async(cols, rows, spp_begin, spp, DEPTH, SEED, 0, args.chunk_spp)
            if use_dist:
                accum.allreduce()
            if rank == 0:
                ctx.resolve(spp_total, out=fb)
            else:
                ctx.sync()

        # one instrumented render: in-kernel ray / termination counters (the work of every later step is identical)
        _, stats = ctx.render(cols, rows, spp_begin, spp, DEPTH, SEED, 0, args.chunk_spp, want_stats=True)
        for _ in range(warmup):
            step()
        barrier()
        trace_ms = []
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
            trace_ms.append(ctx.last_trace_ms())     # HIP events around the trace kernel, on the stream it ran on
        barrier()
        dt = time.perf_counter() - t0
        mine = sum(trace_ms) / len(trace_ms)
        if use_dist:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
            tot = torch.tensor([float(stats["rays"]), float(stats["samples"])], dtype=torch.float64, device="cuda")
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            rays_all, samples_all = float(tot[0].item()), float(tot[1].item())
            every = torch.zeros(world, dtype=torch.float64, device="cuda")
            every[rank] = mine
            dist.all_reduce(every, op=dist.ReduceOp.SUM)
            member_ms = [float(x) for x in every.tolist()]
        else:
            rays_all, samples_all, member_ms = float(stats["rays"]), float(stats["samples"]), [mine]
        return {"dt": dt / steps, "rays": rays_all, "samples": samples_all, "stats": stats, "trace_ms": mine, "member_ms": member_ms, "spp_total": spp_total}

    head = run(weak, args.steps, args.warmup)
    side, exchange_ms = None, None
    if world > 1:
        side = run(not weak, max(1, min(args.steps, 3)), 1)
        ms = []
        for _ in range(4):                               # the exchange alone: ranks synchronised first, wall clock around all-reduce + wait
            barrier()
            t0 = time.perf_counter()
            accum.allreduce()
            ms.append((time.perf_counter() - t0) * 1e3)
        exchange_ms = min(ms[1:])

    if rank == 0:
        stats = head["stats"]
        rehearsal = os.environ.get("RTX_DIST_BACKEND", "nccl") != "nccl"
        out = {
            "metric": "Msamples/s", "value": round(head["samples"] / head["dt"] / 1e6, 3), "unit": "Msamples/s",
            "mrays_per_s": round(head["rays"] / head["dt"] / 1e6, 3),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(head["dt"] * 1e3, 4),
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "book-1 final scene (seed 1, %d spheres) %dx%d, %s, depth %d" % (len(table), cols, rows, ("%d spp per GPU" % args.spp) if weak else ("%d spp in all" % args.spp), DEPTH),
                       "parallelism": (("REHEARSAL: %d ranks over %d device(s), %s all-reduce; not a measurement" % (world, torch.cuda.device_count(), os.environ["RTX_DIST_BACKEND"])) if rehearsal
                                       else "spp split x%d, one process per GPU + RCCL all-reduce (torch.distributed nccl) of the float accumulation buffer" % world) if world > 1 else "1 GPU",
                       "rays_per_sample": round(stats["rays"] / stats["samples"], 4),
                       "candidate_search": "linear list: every ray against every sphere through the f16 MFMA filter (RTX_FILTER_MFMA, the north star's path; the library's default RTX_FILTER_AUTO walks the uniform grid for this table: other_workloads)",
                       "work_order": "cost-ordered from the second launch of the frame on (rtx_last_launch_ordered)"},
            "roofline": roofline_block(len(table), stats, head["trace_ms"]),
        }
        if world > 1:
            out["rccl_ranks"] = world
            out["member_trace_ms"] = [round(m, 4) for m in head["member_ms"]]
            out["exchange"] = {"allreduce_ms": round(exchange_ms, 4), "note": "wall clock around torch.distributed.all_reduce + synchronize after a barrier"}
            out["strong_scaling" if weak else "weak_scaling"] = {
                "value": round(side["samples"] / side["dt"] / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(side["dt"] * 1e3, 4),
                "workload": "%d spp %s" % (side["spp_total"], "in all (strong)" if weak else "= %d per GPU (weak)" % args.spp),
                "member_trace_ms": [round(m, 4) for m in side["member_ms"]]}
        if not args.no_extras and world == 1:
            out["other_workloads"] = other_workloads(ra, local_rank, table)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(table)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    accum.release()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
