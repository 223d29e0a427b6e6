"""Developer experiment: MIOpen find mode / channels-last for the trainer's convolutions."""
import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ct_pvae_amd import trainer as tr
for bench, cl in ((True, False), (True, True)):
    torch.backends.cudnn.benchmark = bench
    args = tr.get_args("--nsa 20 --td 50 -b 5 --ns 2 --api 20 --pnm 1e4 --pnm_start 1e3 --random --normal -i 400 --train".split())
    t = tr.PVAETrainer(args, torch.device("cuda", 0))
    if cl:
        t.enc.to(memory_format=torch.channels_last); t.dec.to(memory_format=torch.channels_last)
        t.input_encode = t.input_encode.contiguous(memory_format=torch.channels_last)
    for _ in range(30): t.train_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): t.train_step()
    torch.cuda.synchronize(); print("cudnn.benchmark", bench, "channels_last", cl, "%.2f ms/step" % ((time.perf_counter() - t0) * 10))
