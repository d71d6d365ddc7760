import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from geobo_amd import hip, _lib
lib = hip.require_gpu()
blocks, iters = 1024, 4000
out = torch.empty(blocks * 256, dtype=torch.float64, device="cuda")
def run(mode, nv):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.geobo_mfma_mix(mode, nv, blocks, 50, C.c_void_p(out.data_ptr()), st), "mix"); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.check(lib.geobo_mfma_mix(mode, nv, blocks, iters, C.c_void_p(out.data_ptr()), st), "mix"); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3
    tf = blocks * 4 * iters * 16 * 2048.0 / t / 1e12
    print("mode %d nv %2d: MFMA %.1f TF/s, VALU ops %.2f Tops/s (x64 lanes)" % (mode, nv, tf, blocks * 4 * iters * 16 * nv * 64 / t / 1e12), flush=True)
for mode, nv in ((0, 0), (1, 2), (1, 4), (1, 8), (1, 16), (2, 4), (2, 8), (2, 16), (2, 32), (3, 8), (3, 16), (3, 32)):
    run(mode, nv)
