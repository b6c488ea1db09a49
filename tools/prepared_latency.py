import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch, bn_amd
from bn_amd import distributed as D
dev = torch.device("cuda", 0)
e = bn_amd.Engine(0); te = D.TorchEngine(e, dev)
N = 1 << 15
P, Q = D.synthetic_points(te, 0, N)
q0 = Q[:1].contiguous()
prep = e.g2_prepare_dev(q0.data_ptr(), 1, te._stream())
Qt = q0.expand(N, 24).contiguous()
out = te.empty(N, 48)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for n in (1, 64, 1024, 2048, 3584, 4096, 8192, 16384, 20000, 32768):
    a = t(lambda: e.pairing_prepared_native_dev(P.data_ptr(), prep, out.data_ptr(), n, stream=te._stream()))
    b = t(lambda: e.pairing_batch_dev(P.data_ptr(), Qt.data_ptr(), out.data_ptr(), n, te._stream()))
    print(f"n={n:6d} native prepared {a:.3f} ms   general pairing_batch {b:.3f} ms")
