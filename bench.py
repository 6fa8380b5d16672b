#!/usr/bin/env python3
"""bench.py -- headline benchmark of the trace loop: book-1 final scene, 1200x800, 500 spp per GPU, depth 50.

One step = one full render of that frame by every rank (weak scaling: each rank traces its own 500-sample range of
every pixel; for N > 1 the accumulation buffers are summed with one RCCL all-reduce per step and rank 0 resolves to
8 bits).  Inputs (sphere table, camera) are resident in HBM before the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--spp S] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

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


def cpu_baseline(table, seconds_budget=20.0):
    """The oracle (a port of the reference CPU path: 40-byte AoS spheres, 64x64 tile jobs, iterative colour) timed on
    this box's host cores on a bounded sample of the same workload: the full 1200x800 frame at `spp` samples."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    cores = max(1, min(16, os.cpu_count() or 1))
    cam = pyoracle.book_camera(COLS, ROWS)
    # calibrate on a strip, then size the sample to the budget
    t0 = time.perf_counter()
    pyoracle.render(table, cam, 40, COLS, 0, 1, DEPTH, SEED, nthreads=cores, tile=40)
    strip = time.perf_counter() - t0
    per_frame_spp = strip * (ROWS / 40.0)
    spp = int(max(1, min(128, seconds_budget // max(per_frame_spp, 1e-3))))
    t0 = time.perf_counter()
    _, _, st = pyoracle.render(table, cam, ROWS, COLS, 0, spp, DEPTH, SEED, nthreads=cores, tile=64)
    dt = time.perf_counter() - t0
    # the same renderer on ONE thread (SURVEY s8(d) asks for both figures): the full view at 1200 x 40, sized to about 3 s
    strip_spp = int(max(4, min(16, 3.0 // max(strip * cores, 1e-3))))
    t1 = time.perf_counter()
    _, _, st1 = pyoracle.render(table, cam, 40, COLS, 0, strip_spp, DEPTH, SEED, nthreads=1, tile=40)
    dt1 = time.perf_counter() - t1
    return {"value": round(st["samples"] / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "single_thread": {"value": round(st1["samples"] / dt1 / 1e6, 4), "sample": "full view at 1200x40, %d spp (%.1f s)" % (strip_spp, dt1)},
            "mrays_per_s": round(st["rays"] / dt / 1e6, 4),
            "sample": "book1 %dx%d, %d spp, depth %d, 64x64 tile jobs on %d threads (%.1f s)" % (COLS, ROWS, spp, DEPTH, cores, dt)}


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
    """Side measurements outside the timed region (one GPU, the faster of two launches after a warm-up, trace-kernel time from HIP events):
    BASELINE configuration 5 (20 164 spheres, 100 spp) with the linear list and with the uniform grid, and the
    bench scene with the grid.  Informational: `value` above is configuration 2 with the linear list."""
    out = {}
    for name, table, spp in (("config5_20164_spheres_100spp", ra.random_scene(SEED, 71), 100), ("config2_book1_500spp", book1, 500)):
        for fname, kind in (("linear_list_mfma", ra.FILTER_MFMA), ("uniform_grid", ra.FILTER_GRID)):
            ctx = ra.Context(device)
            try:
                ctx.set_filter(kind)
                ctx.upload_scene(table)
                ctx.set_camera(ra.book_camera(COLS, ROWS))
                ctx.render(COLS, ROWS, 0, 2, DEPTH, SEED)
                ms = 1e30
                for _ in range(2):
                    _, st = ctx.render(COLS, ROWS, 0, spp, DEPTH, SEED, want_stats=True)
                    ms = min(ms, ctx.last_trace_ms())
                out.setdefault(name, {})[fname] = {"msamples_per_s": round(st["samples"] / ms / 1e3, 1), "kernel_ms": round(ms, 3),
                                                  "exact_tests_per_ray": round(st["candidates"] / max(1, st["rays"]), 2)}
            finally:
                ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=500, help="samples per pixel per GPU")
    ap.add_argument("--cols", type=int, default=COLS)
    ap.add_argument("--rows", type=int, default=ROWS)
    ap.add_argument("--chunk-spp", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed side measurements (configuration 5, grid filter)")
    ap.add_argument("--strong", action="store_true", help="strong scaling (BASELINE's target): --spp is the IMAGE's sample count, split over the ranks by sample index")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group and all-reduce even with one rank (rehearsal of the N > 1 path)")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON: libraries that chat on fd 1 (RCCL prints a version banner there) are
    # sent to stderr for the whole run, and the result goes to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N > 1)" % (args.gpus, world))

    import torch
    import torch.distributed as dist
    import raytracing_amd as ra
    from raytracing_amd.multi_gpu import TorchAccum

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

    cols, rows, spp = args.cols, args.rows, args.spp
    table = ra.random_scene(SEED)
    ctx = ra.Context(local_rank)
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(cols, rows))
    accum = TorchAccum(ctx, cols, rows, torch.device("cuda", local_rank))
    if args.strong:                              # strong scaling: the image has `spp` samples, rank r traces its share of them
        from raytracing_amd.multi_gpu import spp_range
        spp_total = spp
        spp_begin, spp = spp_range(rank, world, 0, spp_total)
        if spp == 0:
            raise SystemExit("bench.py --strong: %d samples cannot feed %d ranks" % (spp_total, world))
    else:                                        # weak scaling: rank r owns global samples [r*spp, (r+1)*spp)
        spp_total = spp * world
        spp_begin = rank * spp

    def step():
        ctx.render_async(cols, rows, spp_begin, spp, DEPTH, SEED, 0, args.chunk_spp)
        if use_dist:
            accum.allreduce()
        if rank == 0:
            ctx.resolve(spp_total)
        else:
            ctx.sync()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    # one instrumented render: in-kernel ray / termination counters (the work of every later step is identical)
    _, stats = ctx.render(cols, rows, spp_begin, spp, DEPTH, SEED, 0, args.chunk_spp, want_stats=True)
    for _ in range(args.warmup):
        step()
    barrier()
    trace_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        trace_ms.append(ctx.last_trace_ms())     # HIP events around the trace kernel, on the stream it ran on
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tot = torch.tensor([float(stats["rays"]), float(stats["samples"])], dtype=torch.float64, device="cuda")
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        rays_all, samples_all = float(tot[0].item()), float(tot[1].item())
    else:
        rays_all, samples_all = float(stats["rays"]), float(stats["samples"])

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        trace_avg_ms = sum(trace_ms) / len(trace_ms)
        tests = stats["sphere_tests"]
        achieved = FLOP_PER_TEST * tests / (trace_avg_ms * 1e-3) / 1e12
        # what the matrix cores actually execute: every ray meets the table padded to 128 spheres, 64 f16-MFMA flop per pair
        n_pad_mfma = (len(table) + 127) // 128 * 128
        mfma_tflops = MFMA_FLOP_PER_TEST * stats["rays"] * n_pad_mfma / (trace_avg_ms * 1e-3) / 1e12
        out = {
            "metric": "Msamples/s", "value": round(samples_all / (dt / args.steps) / 1e6, 3), "unit": "Msamples/s",
            "mrays_per_s": round(rays_all / (dt / args.steps) / 1e6, 3),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("book-1 final scene (seed 1, %d spheres) %dx%d, %d spp in all, depth %d" if args.strong else "book-1 final scene (seed 1, %d spheres) %dx%d, %d spp per GPU, depth %d") % (len(table), cols, rows, spp_total if args.strong else spp, DEPTH),
                       "parallelism": ("spp split x%d + RCCL all-reduce of the float accumulation buffer" % world if os.environ.get("RTX_DIST_BACKEND", "nccl") == "nccl" else "REHEARSAL: %d ranks over %d device(s), %s all-reduce" % (world, torch.cuda.device_count(), os.environ["RTX_DIST_BACKEND"])) if world > 1 else "1 GPU",
                       "rays_per_sample": round(stats["rays"] / stats["samples"], 4)},
            # What the hardware executes for the sphere loop is the f16 MFMA filter: that is the roofline this line is priced
            # against (achieved = filter flop issued on the matrix cores / measured kernel time; dense f16 peak).  The kernel's
            # actual limit is the vector issue port -- sign capture, exact re-tests, samplers and shading share it with the
            # MFMA issue slots -- whose utilisation comes from PMC counters (valu_issue).  BASELINE's north star asks for a
            # fraction of the FP32 VALU peak; that formula (18 flop x rays x n / t) counts arithmetic the filter makes
            # unnecessary and exceeds the peak, so it is reported as a throughput equivalent, not as a fraction.
            "roofline": {"bound": "mfma", "kernel": "trace_kernel", "achieved": round(mfma_tflops, 1), "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(mfma_tflops / F16_MFMA_PEAK_TFLOPS, 4), "traffic": measured_traffic(),
                         "peak_kind": "dense f16 MFMA (MI355X_MICROARCH.md); flop counted: 64 per (ray, padded sphere) pair = 4 v_mfma_f32_32x32x16_f16 per 64 rays x 32 spheres",
                         "kernel_ms": round(trace_avg_ms, 4), "mfma_flop_per_test": MFMA_FLOP_PER_TEST, "padded_spheres": n_pad_mfma,
                         "limiting_resource": "vector issue port (VALU + MFMA issue slots)", "valu_issue": measured_valu_issue(),
                         "sphere_tests_per_launch": tests, "exact_tests_per_launch": stats["candidates"],
                         "fp32_equivalent_tflops": round(achieved, 3),
                         "fp32_equivalent_note": "SURVEY s8(d): 18 flop x (rays x n) / kernel time, against the 157.3 TFLOP/s FP32 vector peak the north star names; not executed arithmetic"},
        }
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
