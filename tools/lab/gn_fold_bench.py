"""Timing of pdr_gn_fold in isolation (back-to-back launches, HIP events)."""
import torch
from point_diffusion_refinement_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B = 32
for tpb, C in [(512, 32), (256, 64), (128, 128), (64, 128), (16, 256), (4, 512), (1, 512)]:
    part = torch.randn(B * tpb, C, 2, device=dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    scale, shift = torch.empty(B, C, device=dev), torch.empty(B, C, device=dev)

    def call():
        _lib.check(lib.pdr_gn_fold(part.data_ptr(), C, tpb, C, 1.0, None, 0, 0, 0, 1.0, B, C, 32, float(tpb * 128), 1e-5,
                                   gamma.data_ptr(), beta.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, 0, None, 0,
                                   torch.cuda.current_stream().cuda_stream), "fold")
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50):
            call()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print("tpb=%4d C=%4d: %.2f us per launch (graph of 50)" % (tpb, C, e0.elapsed_time(e1) / 500 * 1e3))
