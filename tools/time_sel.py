"""Developer tool: the training call's two launches at S = ns * b = 10 objects -- forward + log-likelihood on a 20-angle
subset of the dense 180-angle plan (50 different subsets, replayed from one HIP graph) against the same launch on a
20-angle plan of its own -- and the subset backward (segment kernel) against the planned backward."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ct_pvae_amd import _lib, phantoms
from ct_pvae_amd.forward_functions import RotatePlan
d = torch.device("cuda", 0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(0)
theta = phantoms.dense_theta(180)
dense = RotatePlan(theta, 128, 128, True, d)
small = RotatePlan(theta[::9], 128, 128, True, d)
x = torch.rand((S, 128, 128), device=d)
subs = [torch.from_numpy(rng.permutation(180)[:20].astype(np.int32)).to(d) for _ in range(50)]
mask_d, meas_d = torch.full((S, 180), 0.05, device=d), torch.rand((S, 180, 184), device=d)
mask_s, meas_s = mask_d[:, :20].contiguous(), meas_d[:, :20].contiguous()
pnm = torch.tensor(1e4, device=d)
o, lp, dlp, gx = (torch.empty((S, 20, 184), device=d) for _ in range(3)) + (torch.empty((S, 128, 128), device=d),) if False else (None,) * 4
o, lp, dlp = torch.empty((S, 20, 184), device=d), torch.empty((S, 20, 184), device=d), torch.empty((S, 20, 184), device=d)
gx = torch.empty((S, 128, 128), device=d)
w = torch.ones(S, device=d)


def timed(body, n_per=50):
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (10 * n_per)


def fwd_sel():
    for ai in subs:
        dense.forward_loglik(x, mask_d, meas_d, pnm, 1e-7, out=o, out_lp=lp, out_dlp=dlp, angles_i=ai, dense_inputs=True)

def fwd_sel_same():
    for _ in subs:
        dense.forward_loglik(x, mask_d, meas_d, pnm, 1e-7, out=o, out_lp=lp, out_dlp=dlp, angles_i=subs[0], dense_inputs=True)

def fwd_small():
    for _ in subs:
        small.forward_loglik(x, mask_s, meas_s, pnm, 1e-7, out=o, out_lp=lp, out_dlp=dlp)

def bwd_sel():
    for ai in subs:
        dense.backward(dlp, out=gx, scale=w, angles_i=ai)

subs_h = [ai.cpu() for ai in subs]
def fwd_sel_host():
    for ai in subs_h:
        dense.forward_loglik(x, mask_d, meas_d, pnm, 1e-7, out=o, out_lp=lp, out_dlp=dlp, angles_i=ai, dense_inputs=True)

def fwd_sums_host():
    for ai in subs_h:
        dense.forward_loglik_sums(x, mask_d, meas_d, pnm, 1e-7, angles_i=ai, dense_inputs=True)

def bwd_sel_host():
    for ai in subs_h:
        dense.backward(dlp, out=gx, scale=w, angles_i=ai)

def bwd_small():
    for _ in subs:
        small.backward(dlp, out=gx, scale=w)

for ns in (-1, 1, 2):
    _lib.tune("NS", ns)
    print(f"NS={'auto' if ns < 0 else ns}: fwd+loglik sel (rotating subsets) {timed(fwd_sel):.2f} us, sel (one subset) {timed(fwd_sel_same):.2f} us, "
          f"own 20-angle plan {timed(fwd_small):.2f} us")
_lib.tune("NS")
print(f"host-resident subsets (launch arguments): fwd+loglik {timed(fwd_sel_host):.2f} us, fwd + per-object sums (2 launches) "
      f"{timed(fwd_sums_host):.2f} us, bwd {timed(bwd_sel_host):.2f} us")
for bns in (1, 2):
    for bw in (1, 2, 4):
        _lib.tune("BNS", bns); _lib.tune("BW", bw)
        print(f"  bwd4 sel BNS={bns} BW={bw}: {timed(bwd_sel_host):.2f} us")
_lib.tune("BNS"); _lib.tune("BW")
for g in (1, 2, 3, 4, 5, 6, 8, 10, 12):
    _lib.tune("G", g)
    print(f"G={g}: sel rotating {timed(fwd_sel):.2f} us, own plan {timed(fwd_small):.2f} us")
_lib.tune("G")
print(f"bwd: subset through the segment kernel {timed(bwd_sel):.2f} us, planned backward of a 20-angle plan {timed(bwd_small):.2f} us")
for sn in (1, 2):
    _lib.tune("SEG_NS", sn)
    for ppt in (4, 8):
        _lib.tune("SEG_PPT", ppt)
        print(f"  segment kernel SEG_NS={sn} SEG_PPT={ppt}: {timed(bwd_sel):.2f} us")
