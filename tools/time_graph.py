"""Developer experiment: does replaying the fwd+adj pair from a HIP graph beat launching it (dispatch overhead)?"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import phantoms
from ct_pvae_amd.forward_functions import RotatePlan
dev = torch.device('cuda', 0)
B, N, A = int(sys.argv[1]) if len(sys.argv) > 1 else 50, 128, int(sys.argv[2]) if len(sys.argv) > 2 else 20
theta = phantoms.dense_theta(180)[phantoms.sparse_angle_indices(180, A)] if A < 180 else phantoms.dense_theta(180)
plan = RotatePlan(theta, N, N, True, dev)
x = torch.rand((B, N, N), device=dev); g = torch.rand((B, A, plan.PW), device=dev)
sino = torch.empty((B, A, plan.PW), device=dev); gimg = torch.empty_like(x)
def step():
    plan.forward(x, out=sino); plan.backward(g, out=gimg)
def timeit(f, n=500):
    for _ in range(20): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
print("eager  : %.2f us per step" % timeit(step))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.synchronize()
for k in (1, 10):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(k): step()
    torch.cuda.synchronize()
    print("graph of %2d steps: %.2f us per step" % (k, timeit(gr.replay, 200) / k))
ref_s, ref_g = sino.clone(), gimg.clone()
step(); torch.cuda.synchronize()
print("graph results equal eager:", torch.equal(ref_s, sino), torch.equal(ref_g, gimg))
# two streams: forwards on one, adjoints on the other (independent in this benchmark)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def two_streams(n):
    with torch.cuda.stream(sa):
        for _ in range(n): plan.forward(x, out=sino)
    with torch.cuda.stream(sb):
        for _ in range(n): plan.backward(g, out=gimg)
two_streams(20); torch.cuda.synchronize()
t = time.perf_counter(); two_streams(500); torch.cuda.synchronize()
print("two streams (fwd | adj): %.2f us per step" % ((time.perf_counter() - t) / 500 * 1e6))
# one graph, two branches: 10 forwards on one captured stream and 10 adjoints on another (fork / join inside the capture),
# so that the GPU may run the two independent operators side by side without the host in the loop
gr2 = torch.cuda.CUDAGraph()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.graph(gr2):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        for _ in range(10): plan.forward(x, out=sino)
    with torch.cuda.stream(s2):
        for _ in range(10): plan.backward(g, out=gimg)
    cur.wait_stream(s1); cur.wait_stream(s2)
torch.cuda.synchronize()
print("graph, fwd and adj on parallel branches: %.2f us per (fwd + adj)" % (timeit(gr2.replay, 200) / 10))
step(); torch.cuda.synchronize()
print("parallel-branch results equal eager:", torch.equal(ref_s, sino), torch.equal(ref_g, gimg))
