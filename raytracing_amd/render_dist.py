"""render_dist -- the command line of rtx_cli for one OR several GPUs of a node: samples split over the ranks, periodic
all-reduce of the float accumulation buffers (RCCL), checkpoint / resume, PPM written by rank 0.

    python -m raytracing_amd.render_dist -w 1200 -h 800 -a 500 -m 50 -f out.ppm
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 \\
        -m raytracing_amd.render_dist -w 3840 -h 2160 -a 4096 -m 50 -p 256 -k out.accum -R -f out.ppm

Flags as RayTracing.cpp:42-100 / rtx_cli (-a -b -d -f -m -r; -w -h -s -n; -p -k -R -6); -e N exchanges (and rewrites the
image) every N passes instead of every pass.  BASELINE configuration 3 is the second command.  The checkpoint is the
SUM over ranks, so a job may be resumed on a different number of GPUs.
"""
import argparse
import os
import sys
import time

VARIANTS = {"book1": (11, 0.8, 0.95), "glass": (11, 0.1, 0.95), "scaled": (71, 0.8, 0.95)}   # as rtx_cli -n


def parse(argv):
    ap = argparse.ArgumentParser(add_help=False, description=__doc__)
    ap.add_argument("-a", type=int, default=24)             # NUM_AA_SAMPLES, RayTracing.cpp:31
    ap.add_argument("-b", type=int, default=64)
    ap.add_argument("-d", action="store_true")
    ap.add_argument("-f", default="foo.ppm")
    ap.add_argument("-m", type=int, default=5)              # MAX_RAY_DEPTH, RayTracing.cpp:34
    ap.add_argument("-r", action="store_true")
    ap.add_argument("-w", type=int, default=2000)           # COLS, ROWS, RayTracing.cpp:23-24
    ap.add_argument("-h", type=int, default=1000)
    ap.add_argument("-s", type=int, default=1)
    ap.add_argument("-n", default="book1", choices=sorted(VARIANTS))
    ap.add_argument("-p", type=int, default=0)
    ap.add_argument("-e", type=int, default=1)
    ap.add_argument("-k", default="")
    ap.add_argument("-R", action="store_true")
    ap.add_argument("-6", dest="p6", action="store_true")
    ap.add_argument("-F", default="", choices=["", "valu", "mfma", "grid", "auto"])   # candidate search, as rtx_cli -F (results are identical)
    ap.add_argument("--help", action="help")
    a = ap.parse_args(argv)
    if a.R and not a.k:
        ap.error("-R needs -k <checkpoint file>")
    if a.a <= 0 or a.w <= 0 or a.h <= 0 or a.m < 0:
        ap.error("-a, -w, -h must be positive and -m non-negative")
    return a


def main(argv=None):
    a = parse(sys.argv[1:] if argv is None else argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist
    import raytracing_amd as ra
    from raytracing_amd.multi_gpu import TorchAccum, render_progressive

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.device_count() > 0:
        local_rank %= torch.cuda.device_count()      # a launcher may expose one device per rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # RTX_DIST_BACKEND=gloo: rehearsal on a box with fewer GPUs than ranks (ranks share devices, the sum goes through the
        # host); the product path is "nccl" = RCCL over xGMI
        backend = os.environ.get("RTX_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    cols, rows, spp, depth, seed = a.w, a.h, a.a, a.m, a.s
    flags = ra.FLAG_RECURSIVE if a.r else 0
    table = ra.random_scene(seed, *VARIANTS[a.n])
    ctx = ra.Context(local_rank)
    if a.F:
        ctx.set_filter({"valu": ra.FILTER_VALU, "mfma": ra.FILTER_MFMA, "grid": ra.FILTER_GRID, "auto": ra.FILTER_AUTO}[a.F])
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(cols, rows))
    accum = TorchAccum(ctx, cols, rows, dev)

    done = 0
    if a.R:
        if rank == 0 and os.path.exists(a.k):
            # loads into the bound (zeroed) buffer; a checkpoint of another render is refused, the buffer zeroed again
            try:
                h = ctx.load_accum(a.k)
                if (h["cols"], h["rows"], h["seed"], h["max_depth"], h["flags"]) == (cols, rows, seed, depth, flags) and h["spp_done"] <= spp:
                    done = h["spp_done"]
                    print("render_dist: resuming %s at %d of %d samples" % (a.k, done, spp), flush=True)
                else:
                    print("render_dist: %s belongs to another render; starting over" % a.k, flush=True)
            except ra.RtxError as e:
                print("render_dist: %s; starting over" % e, flush=True)
            if done == 0:
                ctx.use_accum_buffer(cols, rows, accum.t.data_ptr())      # load_accum of another size re-binds nothing, but be sure
                accum.t.zero_()
                torch.cuda.synchronize()
        if world > 1:
            d = torch.tensor([done], dtype=torch.int64, device=dev)
            dist.broadcast(d, src=0)
            done = int(d.item())

    def render_fn(begin, count):
        ctx.render_async(cols, rows, begin, count, depth, seed, flags | ra.FLAG_ACCUMULATE)

    def on_preview(t, samples_done):
        if rank == 0:
            if a.k:
                ctx.use_accum_buffer(cols, rows, t.data_ptr())
                try:
                    ctx.save_accum(a.k, samples_done, seed, depth, flags)
                finally:
                    ctx.use_accum_buffer(cols, rows, accum.t.data_ptr())
            fb = accum.resolve_tensor(t, cols, rows, samples_done, a.b if a.d else 0)
            ra.write_ppm(a.f, cols, rows, fb, binary=a.p6)
            print("pass done: %d of %d samples" % (samples_done, spp), flush=True)
        return True

    t0 = time.perf_counter()
    _, done = render_progressive(render_fn, accum.snapshot, accum.allreduce_tensor, rank, world, spp, a.p, spp_done=done,
                                 preview_every=a.e, on_preview=on_preview)
    ctx.sync()
    if world > 1:
        dist.barrier()
    sec = time.perf_counter() - t0
    if rank == 0:
        print("render_dist: %f s (%d spheres, %d x %d, %d spp, depth %d, %d GPU%s)" % (sec, len(table), cols, rows, done, depth, world, "s" if world > 1 else ""), flush=True)
    accum.release()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
