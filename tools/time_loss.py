"""HIP-event times of the fused image loss (csrc/image_loss.hip), forward and backward launch, as a captured graph of
`reps` forward + backward pairs on one stream (what the fused step replays), plus each direction alone.
usage: [SKGS_HIP_LIB=path] python tools/time_loss.py [W] [H] [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from sk_gs_amd import _C

W = int(sys.argv[1]) if len(sys.argv) > 1 else 800
H = int(sys.argv[2]) if len(sys.argv) > 2 else 800
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lib = _C.load_library()
dev = torch.device('cuda')
g = torch.Generator().manual_seed(0)
pred, gt = torch.rand(3, H, W, generator=g).to(dev), torch.rand(3, H, W, generator=g).to(dev)
nbytes = lib.skgs_image_loss_workspace_bytes(C.c_int32(3), C.c_int32(H), C.c_int32(W))
ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
loss3 = torch.zeros(3, device=dev)
grad = torch.empty_like(pred)


def fwd(stream):
    assert lib.skgs_image_loss_forward(C.c_int32(3), C.c_int32(H), C.c_int32(W), C.c_void_p(pred.data_ptr()), C.c_void_p(gt.data_ptr()),
                                       None, C.c_float(0.8), C.c_float(0.2), None, C.c_void_p(ws.data_ptr()), C.c_size_t(nbytes),
                                       C.c_void_p(stream)) == 0


def bwd(stream):
    assert lib.skgs_image_loss_backward(C.c_int32(3), C.c_int32(H), C.c_int32(W), C.c_void_p(pred.data_ptr()), C.c_void_p(gt.data_ptr()),
                                        None, C.c_float(0.8), C.c_float(0.2), None, C.c_void_p(ws.data_ptr()), C.c_size_t(nbytes),
                                        C.c_void_p(grad.data_ptr()), C.c_void_p(loss3.data_ptr()), C.c_void_p(stream)) == 0


def timed(what):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            what(s.cuda_stream)
        s.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            for _ in range(reps):
                what(s.cuda_stream)
        graph.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(s)
            graph.replay()
            e1.record(s)
            s.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


both = timed(lambda st: (fwd(st), bwd(st)))
print(f'{os.environ.get("SKGS_HIP_LIB", "default")} {W}x{H}: forward {timed(fwd):.1f} us, backward {timed(bwd):.1f} us, pair {both:.1f} us '
      f'(graph of {reps}); loss {loss3.tolist()}')
