"""sha256 of eps (batch 7 with per-face timesteps, batch 64 shared timestep) and of 12 DDPM steps at batch 64: two builds of the
library that claim the same arithmetic must print the same lines.  usage: python tools/eps_hash.py [repo root of the build]"""
import hashlib
import os
import sys

root = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch  # noqa: E402
from hifidiff_amd import sampling, schedulers, synth  # noqa: E402
from hifidiff_amd.refiner import FacialRefiner  # noqa: E402

torch.set_grad_enabled(False)
m = FacialRefiner(16); m.load_state_dict(synth.refiner_state_dict(16)); m.to("cuda:0")
h = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]  # noqa: E731
x, crl, crf = [t.cuda() for t in synth.sample_inputs(7, 16)]
print("eps B=7 per-face ", h(m(x, torch.tensor([980., 500., 0., 37., 861., 250., 999.]), crf, crl).sample))
x, crl, crf = [t.cuda() for t in synth.sample_inputs(64, 16)]
print("eps B=64 t=500   ", h(m(x, 500, crf, crl).sample))
sch = schedulers.DDPMScheduler(clip_sample_range=3.0); sch.timesteps = sch.timesteps[:12]
print("ddpm12 B=64      ", h(sampling.sample(m, x, crf, crl, sch, seed=3)))
