"""Stand-alone probe of the effect that made ops.column_sum necessary: a column sum by ATen's reduction
(x.sum(0, dtype=float32): partial buffer + semaphores for shapes that need a cross-block reduction) inside a recorded hipGraph,
replayed several times with new data.  Prints, per replay, the largest deviation from the eager result.
usage: python tools/dev/aten_sum_graph_repro.py [rows] [cols] [replays]"""
import sys
import torch

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 384
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
w = torch.randn(cols, cols, device=dev, generator=g) / cols ** 0.5
x_static = torch.randn(rows, cols, device=dev, generator=g)


def fn(x):
    y = x @ w                                   # a producer kernel in front of the reduction, as in the model
    return y.sum(0, dtype=torch.float32), y * 1.0001


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        fn(x_static)
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out_sum, out_y = fn(x_static)
worst = 0.0
for i in range(reps):
    x = torch.randn(rows, cols, device=dev, generator=g)
    x_static.copy_(x)
    graph.replay()
    ref_sum, _ = fn(x)
    torch.cuda.synchronize()
    err = float((out_sum - ref_sum).abs().max()) / float(ref_sum.abs().max())
    worst = max(worst, err)
    print(f"replay {i}: max rel deviation of the recorded sum from the eager one {err:.3e}")
print("worst", worst)
