"""Developer tool: random forced cuts of the planned forward's piece list (knobs MIXG_G1 / _G2 / _U1 / _G3 / _U2, both slice pairings) at random
batch sizes and angle counts against one cut for the whole launch (MIXG=0), NaN-poisoned outputs, every bit.   python tools/fuzz_piece_lists.py [cases] [seed]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import _lib
from ct_pvae_amd.forward_functions import RotatePlan
d = torch.device('cuda', 0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
x0 = torch.randn((1600, 128, 128), device=d)
bad = 0
plans = {}
for c in range(cases):
    A = int(rng.choice([3, 7, 20, 33, 60]))
    S = int(rng.integers(130, 1600))
    if A not in plans:
        plans[A] = RotatePlan(rng.uniform(-1.0, 4.0, A), 128, 128, True, d, plan_format="u16")
    plan = plans[A]
    x = x0[:S]
    ns = int(rng.choice([1, 2]))
    units = (S + ns - 1) // ns
    with _lib.tuned("MIXG", 0), _lib.tuned("NS", ns):
        ref = torch.full((S, A, plan.PW), float('nan'), device=d)
        plan.forward(x, out=ref)
    g1, g2, g3 = (int(v) for v in rng.integers(1, 9, 3))
    u1 = int(rng.integers(0, units + 1)); u2 = int(rng.integers(u1, units + 1))
    out = torch.full_like(ref, float('nan'))
    with _lib.tuned("NS", ns), _lib.tuned("MIXG_G1", g1), _lib.tuned("MIXG_G2", g2), _lib.tuned("MIXG_U1", u1), _lib.tuned("MIXG_G3", g3), _lib.tuned("MIXG_U2", u2):
        plan.forward(x, out=out)
    lib = torch.full_like(ref, float('nan'))
    plan.forward(x, out=lib)
    ok = torch.equal(out, ref) and torch.equal(lib, ref)
    if not ok:
        bad += 1
        print(f"DIFFER: S={S} A={A} NS={ns} cut [{u1} x {g1}][{u2 - u1} x {g2}][{units - u2} x {g3}] forced equal {torch.equal(out, ref)} library equal {torch.equal(lib, ref)}", flush=True)
print(f"{cases} random cuts, {bad} differ")
sys.exit(1 if bad else 0)
