"""Does a captured hipMemsetAsync of a few bytes take effect on every replay?  (torch's split reductions zero their semaphores
with one: ATen/native/cuda/Reduce.cuh)"""
import ctypes

import torch

hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = 'cuda:0'
for nbytes in (4, 8, 16, 64, 256, 1024, 4096):
    t = torch.zeros(nbytes // 4, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        t.add_(0)                                                     # a kernel in front
        rc = hip.hipMemsetAsync(t.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        t.add_(1)
    seen = []
    for it in range(4):
        g.replay()
        torch.cuda.synchronize()
        seen.append(t.tolist()[:2])
    print(f'memset of {nbytes} bytes, then += 1, over 4 replays: {seen}')
