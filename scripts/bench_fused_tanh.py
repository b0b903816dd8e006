"""Stand-alone timing of the wide-trunk fusion kernels (dm_linear_tanh, dm_tanh_linear_wgrad, dm_tanh_bwd_colsum) against the
framework ops they replace.  python scripts/bench_fused_tanh.py [B O I]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from deepmimic_mujoco_amd import _lib  # noqa: E402

L = _lib.load_library()
dev = torch.device("cuda", 0)
B, O, I = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 1024, 67)
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
x, w, b = torch.randn(B, I, device=dev), torch.randn(O, I, device=dev) / I ** 0.5, torch.randn(O, device=dev)
y, gy, gz = torch.empty(B, O, device=dev), torch.randn(B, O, device=dev), torch.empty(B, O, device=dev)
dw, db = torch.zeros(O, I, device=dev), torch.zeros(O, device=dev)
O2 = O // 2
y2, gy2, gz2, db2 = torch.rand(B, O2, device=dev), torch.randn(B, O2, device=dev), torch.empty(B, O2, device=dev), torch.zeros(O2, device=dev)


def timeit(name, fn, reps=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-52s %7.1f us" % (name, e0.elapsed_time(e1) * 1000 / reps))


timeit("dm_linear_tanh [%d x %d] <- %d" % (B, O, I), lambda: L.dm_linear_tanh(p(x), p(w), p(b), p(y), B, O, I, st))
timeit("  torch addmm + tanh", lambda: torch.tanh(torch.addmm(b, x, w.t())))
timeit("dm_tanh_linear_wgrad", lambda: L.dm_tanh_linear_wgrad(p(gy), p(y), p(x), p(dw), p(db), B, O, I, st))
timeit("  torch tanh_backward + dm_linear_wgrad", lambda: L.dm_linear_wgrad(p(gy * (1 - y * y)), p(x), p(dw), p(db), B, O, I, st))
timeit("dm_tanh_bwd_colsum [%d x %d]" % (B, O2), lambda: L.dm_tanh_bwd_colsum(p(gy2), p(y2), p(gz2), p(db2), B, O2, st))
timeit("  torch tanh_backward + dm_colsum", lambda: L.dm_colsum(p(torch.ops.aten.tanh_backward(gy2, y2)), B, O2, p(db2), st))
timeit("dm_tanh_bwd_colsum [%d x %d]" % (B, O), lambda: L.dm_tanh_bwd_colsum(p(gy), p(y), p(gz), p(db), B, O, st))
