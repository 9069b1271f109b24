"""Why `hades252_perm_batch` on host memory reaches only 44-49 % of the PCIe ceiling inside a PyTorch process while a native
caller reaches 92-97 %.  One configuration per process (tools/host_path_torch_probe.sh runs the
matrix); prints one line: which libamdhip64 the library is bound to, and the median time of the page-locked and of the
pageable 2^22-state call.

    python tools/host_path_torch_probe.py <label> [--no-torch] [--system-runtime-first]
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
label = sys.argv[1] if len(sys.argv) > 1 else "default"
if "--system-runtime-first" in sys.argv:          # bind the library to /opt/rocm's runtime even though torch comes later
    ctypes.CDLL("/opt/rocm/lib/libamdhip64.so.7", mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(os.path.join(ROOT, "hades252_amd/csrc/libhades252.so"))
if "--no-torch" not in sys.argv:
    import torch
    torch.cuda.init()
    _ = torch.zeros(1, device="cuda")            # the process really is a PyTorch GPU process
if "--system-runtime-first" not in sys.argv:
    if "--no-torch" in sys.argv:
        ctypes.CDLL("/opt/rocm/lib/libamdhip64.so.7", mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(os.path.join(ROOT, "hades252_amd/csrc/libhades252.so"))
lib.hades252_host_alloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
lib.hades252_perm_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.hades252_host_free.argtypes = [ctypes.c_void_p]
maps = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime64" in l})
n = 1 << 22
p = ctypes.c_void_p()
assert lib.hades252_host_alloc(ctypes.byref(p), n * 160) == 0
ctypes.memset(p, 1, n * 160)


def med(fn, reps=7):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        assert fn() == 0
        ts.append(time.perf_counter() - t0)
    return sorted(ts[1:])[len(ts[1:]) // 2] * 1e3


pinned = med(lambda: lib.hades252_perm_batch(p, n))
lib.hades252_host_free(p)
import numpy as np
page = np.ones(20 * n, dtype=np.uint64)
pageable = med(lambda: lib.hades252_perm_batch(page.ctypes.data_as(ctypes.c_void_p), n))
env = " ".join("%s=%s" % (k, os.environ[k]) for k in ("HSA_ENABLE_SDMA", "GPU_MAX_HW_QUEUES", "HSA_ENABLE_INTERRUPT",
                                                       "AMD_DIRECT_DISPATCH", "HIP_FORCE_DEV_KERNARG", "HADES252_HOST_CHUNK") if k in os.environ)
print("%-34s page-locked %7.2f ms = %5.1f GB/s each way   pageable %7.2f ms   [%s] runtimes mapped: %s"
      % (label, pinned, 160 * n / pinned / 1e6, pageable, env or "no env", ", ".join(os.path.relpath(m, "/") for m in maps)),
      flush=True)
