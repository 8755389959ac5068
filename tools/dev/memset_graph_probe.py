"""Does a hipMemsetAsync recorded into a hipGraph run on every replay?  (Follow-up of aten_sum_graph_repro.py: ATen's
cross-block reduction zeroes its semaphores with cudaMemsetAsync at every call.)  usage: python tools/dev/memset_graph_probe.py"""
import ctypes
import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device("cuda:0")
for nbytes in (4, 64, 4096, 1 << 20):
    buf = torch.full((nbytes // 4,), 7, dtype=torch.int32, device=dev)
    out = torch.empty_like(buf)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rc = hip.hipMemsetAsync(buf.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
        out.copy_(buf)
        buf.add_(1)                                   # leave something non-zero behind for the next replay
    res = []
    for i in range(4):
        g.replay()
        torch.cuda.synchronize()
        res.append(int(out.abs().max()))
    print(f"{nbytes:8d} bytes: hipMemsetAsync rc {rc}; max |value| seen behind the memset in replays 0..3: {res}  (0 = the memset ran)")
