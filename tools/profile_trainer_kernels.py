"""Developer check: device-side launch census of the steady-state P-VAE training step (config 3)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from torch.profiler import profile, ProfilerActivity
from ct_pvae_amd import trainer as tr
args = tr.get_args("--nsa 20 --td 50 -b 5 --ns 2 --api 20 --pnm 1e4 --pnm_start 1e3 --random --normal -i 400 --train --miopen_find".split())
t = tr.PVAETrainer(args, torch.device("cuda", 0))
for _ in range(30): t.train_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): t.train_step()
torch.cuda.synchronize()
print(f"step {(time.perf_counter() - t0) / 50 * 1e3:.2f} ms")
N = 10
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(N): t.train_step()
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages() if e.device_time_total > 0 and e.device_type.name != "CPU"] or prof.key_averages()
rows = sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print(f"device time per step {tot / N / 1e3:.2f} ms")
print("--- by device time")
for e in rows[:35]:
    if e.self_device_time_total > 0:
        print(f"{e.self_device_time_total / N:9.1f} us/step {e.count / N:7.1f} calls/step  {e.key[:100]}")
print("--- CPU ops by count")
for e in sorted(prof.key_averages(), key=lambda e: -e.count)[:40]:
    print(f"{e.count / N:7.1f}/step  self cpu {e.self_cpu_time_total / N:8.1f} us/step  {e.key[:90]}")
