"""Pageable host <-> device copies by SIZE (resident pages): what a chunked upload / download pays per call.
hipMemcpyAsync on a non-blocking stream followed by a stream synchronisation, best of 5."""
import ctypes as C, time, sys
import numpy as np
hip = C.CDLL("libamdhip64.so")
def ck(e):
    if e: raise RuntimeError("hip error %d" % e)
ck(hip.hipSetDevice(0))
cap = 512 << 20
d = C.c_void_p(); ck(hip.hipMalloc(C.byref(d), C.c_size_t(cap)))
st = C.c_void_p(); ck(hip.hipStreamCreateWithFlags(C.byref(st), 1))
host = np.ones(cap, np.uint8)
for mb in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
    b = mb << 20
    best = [1e9, 1e9, 1e9]
    for rep in range(5):
        t0 = time.perf_counter(); ck(hip.hipMemcpyAsync(d, C.c_void_p(host.ctypes.data), C.c_size_t(b), 1, st)); t1 = time.perf_counter(); ck(hip.hipStreamSynchronize(st)); t2 = time.perf_counter()
        best[0] = min(best[0], t2 - t0); best[2] = min(best[2], t1 - t0)
        t0 = time.perf_counter(); ck(hip.hipMemcpyAsync(C.c_void_p(host.ctypes.data), d, C.c_size_t(b), 2, st)); ck(hip.hipStreamSynchronize(st)); t2 = time.perf_counter()
        best[1] = min(best[1], t2 - t0)
    print("%4d MB: H2D %7.3f ms (%5.1f GB/s; the call itself returns after %7.3f ms) | D2H %7.3f ms (%5.1f GB/s)" % (mb, best[0] * 1e3, b / best[0] / 1e9, best[2] * 1e3, best[1] * 1e3, b / best[1] / 1e9), flush=True)
# fresh (never touched) destination, 4 KB pages against a madvise(MADV_HUGEPAGE) range
import mmap
libc = C.CDLL("libc.so.6")
for huge in (0, 1):
    b = 128 << 20
    m = mmap.mmap(-1, b + (2 << 20))
    addr = C.addressof(C.c_char.from_buffer(m)); a2 = (addr + (2 << 20) - 1) & ~((2 << 20) - 1)
    if huge: libc.madvise(C.c_void_p(a2), C.c_size_t(b), 14)
    t0 = time.perf_counter(); ck(hip.hipMemcpyAsync(C.c_void_p(a2), d, C.c_size_t(b), 2, st)); ck(hip.hipStreamSynchronize(st)); t = time.perf_counter() - t0
    print("128 MB D2H into FRESH pages (%s): %.2f ms" % ("huge pages asked for" if huge else "4 KB pages", t * 1e3))
