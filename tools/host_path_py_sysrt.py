# the host path from a Python process that has NOT loaded torch: ctypes over the system HIP runtime
import ctypes, time, sys, os
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so", mode=ctypes.RTLD_GLOBAL)
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hades252_amd/csrc/libhades252.so"))
lib.hades252_host_alloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
lib.hades252_perm_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.hades252_host_free.argtypes = [ctypes.c_void_p]
for logn in (22, 24):
    n = 1 << logn
    p = ctypes.c_void_p()
    assert lib.hades252_host_alloc(ctypes.byref(p), n * 160) == 0
    ctypes.memset(p, 1, n * 160)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); rc = lib.hades252_perm_batch(p, n); ts.append(time.perf_counter() - t0)
        assert rc == 0
    dt = sorted(ts[1:])[2]
    print("python + ctypes, system HIP runtime, no torch: n=2^%d %.3f ms %.1f Mperm/s %.2f GB/s each way" % (logn, dt * 1e3, n / dt / 1e6, 160 * n / dt / 1e9))
    lib.hades252_host_free(p)
