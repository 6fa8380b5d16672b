"""Diagnostic: is the batch list sorted by strip cost?"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra
cols, rows, spp = 1200, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 62
ctx = ra.Context(0); ctx.upload_scene(ra.random_scene(1)); ctx.set_camera(ra.book_camera(cols, rows))
for k in range(3):
    ctx.render(cols, rows, 0, spp, 50, 1)
L = ra.lib()
cost = np.zeros(1 << 20, dtype=np.uint32); order = np.zeros(1 << 24, dtype=np.uint32)
nc, no = C.c_size_t(), C.c_size_t()
L.rtx_debug_order.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
rc = L.rtx_debug_order(ctx._h, cost.ctypes.data, cost.size, order.ctypes.data, order.size, C.byref(nc), C.byref(no))
cost, order = cost[:nc.value], order[:no.value]
print("rc", rc, "strips", len(cost), "batches", len(order), "plan", ctx.last_chunk_plan())
print("cost: min %d max %d mean %.1f; sky strips (cost == spp): %d" % (cost.min(), cost.max(), cost.mean(), (cost == spp).sum()))
strip, chunk = order & 0x3FFFFF, order >> 22
print("first 24 batches (strip, chunk, cost):", [(int(s), int(c), int(cost[s])) for s, c in zip(strip[:24], chunk[:24])])
print("last 8:", [(int(s), int(c), int(cost[s])) for s, c in zip(strip[-8:], chunk[-8:])])
lc = cost[strip].astype(np.int64)
print("list is non-increasing in strip cost:", bool((np.diff(lc) <= 0).all()), " every (strip, chunk) once:", len(np.unique(order)) == len(order))
