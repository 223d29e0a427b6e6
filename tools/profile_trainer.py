"""Developer check: host-side profile of one P-VAE training step (config 3)."""
import cProfile, io, os, pstats, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import trainer as tr
args = tr.get_args("--nsa 20 --td 50 -b 5 --ns 2 --api 20 --pnm 1e4 --pnm_start 1e3 --random --normal -i 200 --train".split())
t = tr.PVAETrainer(args, torch.device("cuda", 0))
for _ in range(10): t.train_step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(50): t.train_step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(30); print(s.getvalue()[:5000])
