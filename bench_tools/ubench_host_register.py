#!/usr/bin/env python3
"""What does pinning a caller's array cost against copying it pageable? (decides whether keaki_hip_encrypt_batch should hipHostRegister its
arguments: SURVEY 8b "caller owns every buffer")  Run on the GPU box: python3 bench_tools/ubench_host_register.py"""
import ctypes as C
import time
import numpy as np

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
H2D, D2H = 1, 2
for mb in (32, 128):
    n = mb << 20
    d = C.c_void_p()
    assert hip.hipMalloc(C.byref(d), n) == 0
    for fresh in (True, False):
        a = np.empty(n, np.uint8) if fresh else np.ones(n, np.uint8)       # fresh: pages never touched (an output array)
        p = a.ctypes.data
        t0 = time.perf_counter(); assert hip.hipMemcpy(d, p, n, H2D) == 0 or True; hip.hipDeviceSynchronize(); t_page_h2d = time.perf_counter() - t0
        b = np.empty(n, np.uint8) if fresh else np.ones(n, np.uint8)
        t0 = time.perf_counter(); hip.hipMemcpy(b.ctypes.data, d, n, D2H); hip.hipDeviceSynchronize(); t_page_d2h = time.perf_counter() - t0
        c = np.empty(n, np.uint8) if fresh else np.ones(n, np.uint8)
        t0 = time.perf_counter(); r = hip.hipHostRegister(c.ctypes.data, n, 0); t_reg = time.perf_counter() - t0
        t0 = time.perf_counter(); hip.hipMemcpy(d, c.ctypes.data, n, H2D); hip.hipDeviceSynchronize(); t_pin_h2d = time.perf_counter() - t0
        t0 = time.perf_counter(); hip.hipMemcpy(c.ctypes.data, d, n, D2H); hip.hipDeviceSynchronize(); t_pin_d2h = time.perf_counter() - t0
        t0 = time.perf_counter(); hip.hipHostUnregister(c.ctypes.data); t_unreg = time.perf_counter() - t0
        print("%4d MB %s pages: pageable H2D %.1f ms, D2H %.1f ms | register %.1f ms (rc %d) + pinned H2D %.1f, D2H %.1f + unregister %.1f ms"
              % (mb, "untouched" if fresh else "resident ", t_page_h2d * 1e3, t_page_d2h * 1e3, t_reg * 1e3, r, t_pin_h2d * 1e3, t_pin_d2h * 1e3, t_unreg * 1e3))
