import ctypes, mmap, time, numpy as np
print(open('/sys/kernel/mm/transparent_hugepage/enabled').read().strip(), '|', open('/sys/kernel/mm/transparent_hugepage/defrag').read().strip())
libc = ctypes.CDLL(None, use_errno=True)
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
n = 160 << 20
def touch(a):
    t0 = time.perf_counter(); a[::4096] = 1; return (time.perf_counter() - t0) * 1e3
a = np.zeros(n, np.uint8); print("plain np.zeros touch: %.1f ms" % touch(a))
b = np.zeros(n, np.uint8)
p = b.ctypes.data; lo = (p + (2 << 20) - 1) & ~((2 << 20) - 1); ln = (p + n - lo) & ~((2 << 20) - 1)
r = libc.madvise(lo, ln, 14)   # MADV_HUGEPAGE
print("madvise rc", r, ctypes.get_errno()); print("madvise(HUGEPAGE) touch: %.1f ms" % touch(b))
c = np.empty(n, np.uint8); print("np.empty touch: %.1f ms" % touch(c))
t0=time.perf_counter(); d = np.zeros(n, np.uint8); print("np.zeros alloc %.2f ms" % ((time.perf_counter()-t0)*1e3))
