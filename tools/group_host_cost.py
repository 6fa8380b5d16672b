"""Host cost of driving several members from one thread: rtx_group_render_frame on a tiny frame (GPU work negligible) with 1, 2, 4, 8
members in loopback (RTX_GROUP_LOOPBACK=1).  python tools/group_host_cost.py"""
import os, sys, time
os.environ["RTX_GROUP_LOOPBACK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ctypes as C
import raytracing_amd as ra
ra.lib().rtx_debug_group_enqueue_us.restype = C.c_double
ra.lib().rtx_debug_group_enqueue_us.argtypes = [C.c_void_p]
table = ra.random_scene(1)
cols, rows = 64, 64
for n in (1, 2, 4, 8):
    g = ra.Group([0] * n); g.upload_scene(table); g.set_camera(ra.book_camera(cols, rows))
    fb = np.zeros(cols * rows, dtype=np.uint32)
    for _ in range(5):
        g.render_frame(cols, rows, 0, 8 * n, 5, 1, 8 * n, out=fb)
    t0 = time.perf_counter(); K = 50; enq = []
    for _ in range(K):
        g.render_frame(cols, rows, 0, 8 * n, 5, 1, 8 * n, out=fb)
        enq.append(ra.lib().rtx_debug_group_enqueue_us(g._h))
    print("members %d: %.3f ms per frame; queueing the members' launches took %.0f us (median; %s)" % (n, (time.perf_counter() - t0) / K * 1e3, sorted(enq)[K // 2], "serial" if os.environ.get("RTX_GROUP_SERIAL") else "one host thread per member"), flush=True)
    g.close()
