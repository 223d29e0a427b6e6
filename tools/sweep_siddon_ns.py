"""Developer tool: the TomoPy-style forward over slices per walk (knob SIDDON_NS: 1 / 2 = the LDS kernels, 4 / 8 = packed objects through the L2) against
the library's choice, bit-compared."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ct_pvae_amd import _lib, phantoms
from ct_pvae_amd.helper_functions import create_sinograms
from time_modes import graph_time
dev = torch.device('cuda', 0)
for B, A in ((3, 20), (5, 20), (8, 20), (16, 20), (24, 20), (32, 20), (50, 20), (3, 180), (4, 180), (8, 90), (50, 180)):
    x = torch.rand((B, 128, 128), device=dev); theta = phantoms.dense_theta(A)
    ref = create_sinograms(x, theta)
    for _ in range(2): graph_time(lambda: create_sinograms(x, theta), 20)
    lib = min(graph_time(lambda: create_sinograms(x, theta), 20) for _ in range(3)) * 1e6
    row = []
    for ns in (1, 2, 4, 8):
        with _lib.tuned("SIDDON_NS", ns):
            out = create_sinograms(x, theta)
            t = min(graph_time(lambda: create_sinograms(x, theta), 20) for _ in range(3)) * 1e6
        row.append(f"NS={ns} {t:.1f}{'' if torch.equal(out, ref) else ' DIFFER'}")
    print(f"B={B} A={A}: library {lib:.1f} us | " + " ; ".join(row), flush=True)
