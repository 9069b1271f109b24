import sys, time
sys.path.insert(0, "/root/repo")
import torch
from hades252_amd import strategy as H
n = (1 << 30) + 1000
print("free/total GB:", [x / 1e9 for x in torch.cuda.mem_get_info()])
a = H.gen_b(5 * n, "cuda")
torch.cuda.synchronize(); t = time.time()
H.ScalarStrategy().perm(a)
torch.cuda.synchronize(); dt = time.time() - t
d1 = H.digest(a)
print("one call n=2^30+1000: %.3f s  %.1f Mperm/s" % (dt, n / dt / 1e6), ["%016x" % x for x in d1])
H.gen_b(5 * n, "cuda", out=a)
flat = a.view(-1)
cut = 20 * ((1 << 29) + 12345)
H.ScalarStrategy().perm(flat[:cut]); H.ScalarStrategy().perm(flat[cut:])
d2 = H.digest(a)
print("two calls agree:", d1 == d2)
