import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
def chk(r):
    assert r == 0, r
for gb in (8, 16, 32, 40):
    n = gb * (1 << 30)
    p = C.c_void_p()
    t0 = time.perf_counter(); chk(hip.hipMalloc(C.byref(p), C.c_size_t(n))); t1 = time.perf_counter()
    chk(hip.hipMemsetAsync(p, 0, C.c_size_t(n), None)); chk(hip.hipDeviceSynchronize()); t2 = time.perf_counter()
    chk(hip.hipMemsetAsync(p, 0, C.c_size_t(n), None)); chk(hip.hipDeviceSynchronize()); t3 = time.perf_counter()
    chk(hip.hipFree(p)); t4 = time.perf_counter()
    print(f"{gb} GB: malloc {t1-t0:.3f}s memset#1 {t2-t1:.3f}s memset#2 {t3-t2:.3f}s free {t4-t3:.3f}s")
