"""How long the reference-format activation cache takes to write (12 layers x [10000, 768] fp32 + embeddings), piece by
piece, and whether threads help: device transpose + device->host copy, torch.save."""
import os, sys, time, tempfile, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
U, N, L = 768, 10000, 12
At = torch.randn(L * U, 10048, device=dev)
tmp = tempfile.mkdtemp(prefix="mcd_cw_")
torch.cuda.synchronize()

def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, time.perf_counter() - t0

host, dt = t(lambda: [At[i * U:(i + 1) * U, :N].t().contiguous().cpu() for i in range(L)])
print("transpose + pageable D2H, 12 layers: %.1f ms" % (dt * 1e3))
pin = [torch.empty((N, U), pin_memory=True) for _ in range(L)]
_, dt = t(lambda: [pin[i].copy_(At[i * U:(i + 1) * U, :N].t(), non_blocking=True) for i in range(L)])
print("transpose + pinned D2H, 12 layers:   %.1f ms" % (dt * 1e3))
_, dt = t(lambda: [torch.save(host[i], os.path.join(tmp, "a%d.pt" % i)) for i in range(L)])
print("torch.save x12 serial:               %.1f ms" % (dt * 1e3))
def par(n):
    ths = [threading.Thread(target=lambda i=i: [torch.save(host[j], os.path.join(tmp, "b%d.pt" % j)) for j in range(i, L, n)]) for i in range(n)]
    [x.start() for x in ths]; [x.join() for x in ths]
for n in (2, 4, 6, 12):
    _, dt = t(lambda: par(n))
    print("torch.save x12 in %2d threads:        %.1f ms" % (n, dt * 1e3))
import numpy as np
_, dt = t(lambda: [np.save(os.path.join(tmp, "c%d.npy" % i), host[i].numpy()) for i in range(L)])
print("np.save x12 serial (for scale):      %.1f ms" % (dt * 1e3))
print("tmp dir:", tmp, os.statvfs(tmp).f_bsize)
