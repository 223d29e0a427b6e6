"""P-VAE training harness around the HIP projector (BASELINE config 3/4).

The reference's driver is ctvae/main_ct_vae.py (CT_VAE.train / train_step, :375-486) with the ELBO of
ctvae/helper_functions.py:204-332 and the nets of ctvae/models.py.  The nets and the driver are host code: here they
are plain PyTorch-ROCm (MIOpen convs); the physics decoder -- calculate_log_prob_M_given_R, the only differentiated
caller of the projector -- runs on the hand-written kernels.  What is reproduced from the reference, line by line:

  * `ns` decoder samples per step, each projected through `api` angles drawn per step from a shuffled stream of
    0..A-1 (ctvae/helper_functions.py:104-107, :263-295)
  * TruncatedNormal(positive_range(alpha), positive_range(beta), low=0, high=1e10) output distribution (:273)
  * log p(M|R): Normal(loc=proj*mask, scale=eps+sqrt(loc/pnm+eps)) (:360-368); pnm annealed as
    pnm * factor**iter with factor = exp(log(pnm/pnm_start)/num_iter) (ctvae/main_ct_vae.py:146-149, :392 -- the
    variable already holds the FINAL pnm, so the effective multiplier runs pnm .. pnm^2/pnm_start; kept as is)
  * loss = mean over the batch / 1e5 (:478); NaN gradients zeroed, per-tensor clip_by_norm(100), Adam(1e-4, eps 1e-7)
    (:353, :482-485); encoder input scaled by 1/300 (ctvae/helper_functions.py:239)
  * NaN loss stops the run (:401-402); checkpoint every save_interval and at the end (:409-415)

Data parallelism is new (the reference is single-device): every rank trains on its shard of the batch and the flat
gradient bucket (~0.7 M fp32) is summed with ONE all-reduce per step (RCCL over xGMI; gloo in the CPU tests).

Synthetic data stands in for the reference's dataset files: foam phantoms (phantoms.py) -> dense sinograms with the
TomoPy-style projector on the GPU (create_sinograms) -> dose masks and Poisson noise (ctvae/create_masks.py:45-63,
:94-95) -> initial reconstructions for the encoder by FBP on the GPU (iradon) in place of tomopy.recon
(ctvae/helper_functions.py:477-529): one ramp-filtered channel plus the unfiltered back-projection of the mask.
"""
import argparse
import math
import os
import time

import numpy as np
import torch
import torch.nn as nn

from . import dataset_io, phantoms, sharding
from .create_masks import create_all_masks
from .fbp import iradon_all
from .forward_functions import num_proj_pix
from .helper_functions import calculate_log_prob_M_given_R, create_sinograms

EPS32 = float(np.finfo(np.float32).eps)


# ---------------------------------------------------------------------------------------------------------
# nets (ctvae/models.py), channels-first
# ---------------------------------------------------------------------------------------------------------
def positive_range(x, offset=EPS32):
    """ctvae/helper_functions.py:198-201"""
    x = x - 1
    neg = (x < 0).to(x.dtype)
    return (torch.exp(torch.clamp(x, -1e10, 10)) + offset) * neg + (x + 1) * (1 - neg)


class _Maxout(torch.autograd.Function):
    """max of the two halves of the channel axis (ctvae/models.py:330-341, tf.maximum of two convolutions).  One
    comparison mask kept for the backward, which writes both halves of the gradient directly: 2 + 3 launches instead of
    the 14 of chunk + torch.maximum under autograd (two slice-backward fills and copies, mask, tie handling, add).
    Ties send the gradient to the first convolution, as TensorFlow's MaximumGrad does (x >= y)."""

    @staticmethod
    def forward(ctx, y):
        c = y.shape[1] // 2
        a, b = y[:, :c], y[:, c:]
        first = a >= b
        ctx.save_for_backward(first)
        return torch.where(first, a, b)

    @staticmethod
    def backward(ctx, g):
        first, = ctx.saved_tensors
        c = g.shape[1]
        gy = g.new_empty((g.shape[0], 2 * c) + tuple(g.shape[2:]))
        torch.mul(g, first, out=gy[:, :c])
        torch.sub(g, gy[:, :c], out=gy[:, c:])        # g where the second convolution won, 0 elsewhere (exact)
        return gy


class _PeriodicPad(torch.autograd.Function):
    """'periodic' padding of the two spatial axes (ctvae/models.py:219-263) as two gathers; the backward is the two
    matching scatter-adds (every source pixel receives at most two terms per axis, so the sum does not depend on the
    order).  4 + 6 launches instead of the ~20 of F.pad(mode='circular') under autograd (slice assignments forward,
    slice-backward fills, copies and adds backward)."""
    _index = {}

    @classmethod
    def index(cls, n, lo, hi, device):
        key = (n, lo, hi, str(device))
        idx = cls._index.get(key)
        if idx is None:
            idx = cls._index[key] = (torch.arange(-lo, n + hi, device=device) % n)
        return idx

    @staticmethod
    def forward(ctx, x, pads):
        wl, wr, hl, hr = pads
        ctx.ih = _PeriodicPad.index(x.shape[2], hl, hr, x.device)
        ctx.iw = _PeriodicPad.index(x.shape[3], wl, wr, x.device)
        ctx.hw = x.shape[2:]
        return x.index_select(2, ctx.ih).index_select(3, ctx.iw)

    @staticmethod
    def backward(ctx, g):
        H, W = ctx.hw
        gw = g.new_zeros(g.shape[:3] + (W,)).index_add_(3, ctx.iw, g)
        return gw.new_zeros(gw.shape[:2] + (H, W)).index_add_(2, ctx.ih, gw), None


class ConvBlock(nn.Module):
    """Conv2D => maxout of two convolutions (ctvae/models.py:267-342); 'periodic' padding for the strided/plain
    convolutions (:219-263), Conv2DTranspose(padding='same') for up-sampling."""

    def __init__(self, cin, cout, k, stride, transpose):
        super().__init__()
        self.k, self.stride, self.transpose = k, stride, transpose
        # the two convolutions of the maxout are ONE convolution with 2 * cout output channels (half the launches);
        # each half is initialised as its own GlorotUniform layer
        if transpose:
            pad = (k - stride + 1) // 2
            opad = stride + 2 * pad - k
            if not 0 <= opad < stride:             # e.g. the toy recipe's --ks 2 --se 1: one pixel too many, cropped by the decoder
                pad = max((k - stride) // 2, 0)
                opad = min(max(stride + 2 * pad - k, 0), stride - 1)
            self.ab = nn.ConvTranspose2d(cin, 2 * cout, k, stride=stride, padding=pad, output_padding=opad)
            halves = self.ab.weight.data.chunk(2, dim=1)        # ConvTranspose2d weight: [cin][cout][k][k]
        else:
            self.ab = nn.Conv2d(cin, 2 * cout, k, stride=stride, padding=0)
            halves = self.ab.weight.data.chunk(2, dim=0)        # Conv2d weight: [cout][cin][k][k]
        for h in halves:
            nn.init.xavier_uniform_(h)             # GlorotUniform (fan-in / fan-out of ONE of the two convolutions)
        nn.init.zeros_(self.ab.bias)

    def forward(self, x):
        if not self.transpose:
            pads = []
            for n in (x.shape[-1], x.shape[-2]):   # last axis first, as torch.nn.functional.pad orders them
                p = self.k - (n % self.stride if n % self.stride else self.stride)
                pads += [p // 2 + p % 2, p // 2]
            x = _PeriodicPad.apply(x, tuple(pads))
        return _Maxout.apply(self.ab(x))


class EncodeNet(nn.Module):
    """create_encode_net, ctvae/models.py:23-108: returns the list of skips (the first one is the repeated input)."""

    def __init__(self, cin, feature_maps, fmm, kernel, stride, inter_layers, inter_kernel):
        super().__init__()
        self.fmm = fmm
        c = cin * fmm
        self.channels = [c]
        self.blocks = nn.ModuleList()
        for f in feature_maps:
            layers = [ConvBlock(c, c, inter_kernel, 1, False) for _ in range(inter_layers)]
            layers.append(ConvBlock(c, f * fmm, kernel, stride, False))
            self.blocks.append(nn.Sequential(*layers))
            c = f * fmm
            self.channels.append(c)

    def forward(self, x):
        x = x.repeat_interleave(self.fmm, dim=1)   # tf.repeat(output, feature_maps_multiplier, axis=-1)
        skips = [x]
        for blk in self.blocks:
            x = blk(x)
            skips.append(x)
        return skips


class DecodeNet(nn.Module):
    """create_decode_net, ctvae/models.py:112-215 (the code concatenates every skip, including the input level)."""

    def __init__(self, enc_channels, fmm, out_channels, kernel, stride, inter_layers, inter_kernel):
        super().__init__()
        lat = [c // fmm for c in enc_channels]     # channels of the sampled latents
        self.ups = nn.ModuleList()
        c = lat[-1]
        for lvl in range(len(enc_channels) - 2, -1, -1):
            layers = [ConvBlock(c, enc_channels[lvl], kernel, stride, True)]
            layers += [ConvBlock(enc_channels[lvl], enc_channels[lvl], inter_kernel, 1, False) for _ in range(inter_layers)]
            self.ups.append(nn.Sequential(*layers))
            c = enc_channels[lvl] + lat[lvl]
        self.head = ConvBlock(c, 2 * out_channels, kernel, 1, False)

    def forward(self, latents):
        x = latents[-1]
        for up, skip in zip(self.ups, reversed(latents[:-1])):
            x = up(x)
            dx, dy = x.shape[-2] - skip.shape[-2], x.shape[-1] - skip.shape[-1]
            x0, y0 = dx // 2 + dx % 2, dy // 2 + dy % 2
            x = x[..., x0:x0 + skip.shape[-2], y0:y0 + skip.shape[-1]]
            x = torch.cat([x, skip], dim=1)
        alpha, beta = self.head(x).chunk(2, dim=1)
        return alpha, beta


# ---------------------------------------------------------------------------------------------------------
# distributions
# ---------------------------------------------------------------------------------------------------------
_SQRT2 = math.sqrt(2.0)


def _ncdf(z):
    return 0.5 * (1 + torch.erf(z / _SQRT2))


class TruncatedNormal:
    """tfd.TruncatedNormal(loc, scale, low, high): reparameterised sample by inverse CDF, log_prob."""

    def __init__(self, loc, scale, low=0.0, high=1e10):
        self.loc, self.scale = loc, scale
        self.a, self.b = (low - loc) / scale, (high - loc) / scale
        self.cdf_a, self.cdf_b = _ncdf(self.a), _ncdf(self.b)
        self.Z = (self.cdf_b - self.cdf_a).clamp_min(1e-30)

    def rsample(self):
        u = torch.rand_like(self.loc)
        p = (self.cdf_a + u * self.Z).clamp(1e-7, 1 - 1e-7)
        z = torch.special.ndtri(p)
        return (self.loc + self.scale * z).clamp_min(0.0)

    def log_prob(self, x):
        z = (x - self.loc) / self.scale
        return -0.5 * z * z - 0.5 * math.log(2 * math.pi) - torch.log(self.scale) - torch.log(self.Z)


def kl_normal_std(loc, scale):
    """KL(N(loc, scale) || N(0, 1)), elementwise."""
    return 0.5 * (scale * scale + loc * loc - 1.0) - torch.log(scale)


# ---------------------------------------------------------------------------------------------------------
# ELBO (ctvae/helper_functions.py:204-332; --normal: Normal latents + TruncatedNormal output, otherwise the reference's
# default Beta latents, Beta(0.5, 0.5) prior and Beta output, :247-252, :275-285, ctvae/main_ct_vae.py:369-372)
# ---------------------------------------------------------------------------------------------------------
def find_loss_vae_unsup(proj_sample, mask, input_encode, model_encode, model_decode, poisson_noise_multiplier, sqrt_reg,
                        kl_anneal, kl_multiplier, num_samples=2, theta=None, angles_i=None, pad=True, deterministic=False,
                        use_normal=True):
    skips = model_encode(input_encode / 300)
    q = None
    if not deterministic:
        q = []
        for sk in skips:
            loc, log_scale = sk.chunk(2, dim=1)
            if use_normal:
                q.append((loc, positive_range(log_scale) + sqrt_reg))
            else:                                                   # :252 tfd.Beta(positive_range(loc), scale)
                q.append(torch.distributions.Beta(positive_range(loc), positive_range(log_scale)))
    # The reference draws its `num_samples` latent samples in a Python loop (:263-312); they are independent, so here
    # they ride the batch axis: ONE decoder pass and ONE projector + likelihood pass over num_samples * B objects
    # (sample-major), the same estimator with half the launches at ns = 2.
    ns = 1 if deterministic else int(num_samples)
    B = input_encode.shape[0]
    if deterministic:
        q_sample = skips
    elif use_normal:
        q_sample = [loc.repeat(ns, 1, 1, 1) + scale.repeat(ns, 1, 1, 1) * torch.randn((ns * B,) + tuple(loc.shape[1:]),
                                                                                      device=loc.device, dtype=loc.dtype)
                    for loc, scale in q]
    else:                                                           # reparameterised Beta samples, sample-major
        q_sample = [d.rsample((ns,)).reshape((ns * B,) + tuple(d.concentration1.shape[1:])) for d in q]
    alpha, beta = model_decode(q_sample)
    if use_normal:
        dist = TruncatedNormal(positive_range(alpha), positive_range(beta), low=0.0, high=1e10)
        output_sample = dist.rsample()                                   # [ns * B][1][X][Y]
        log_prob_R_given_z = dist.log_prob(output_sample)
    else:                                                           # :278-285
        dist = torch.distributions.Beta(positive_range(alpha), positive_range(beta))
        output_sample = dist.rsample()
        log_prob_R_given_z = dist.log_prob(output_sample.clamp(sqrt_reg, 1 - sqrt_reg))
    # per-object sums of the log-probabilities, reduced inside the projector launch
    lp = calculate_log_prob_M_given_R(output_sample.permute(0, 2, 3, 1), mask.repeat(ns, 1), proj_sample.repeat(ns, 1, 1),
                                      poisson_noise_multiplier, sqrt_reg, theta=theta, angles_i=angles_i, pad=pad,
                                      reduce="per_object")
    # :305-306 reduce_sum(..., axis=[0, 1, 2]) of the squeezed [B][A][P] and [B][X][Y] tensors: the log-likelihood of a
    # sample is ONE number for the whole batch (the batch axis is summed too), the KL below is per object; :329-330 then
    # broadcast-subtract, and train_step takes the mean over the batch -- i.e. mean_b(KL_b) - sum_b(loglik_b).
    log_prob_M = (lp + log_prob_R_given_z.sum(dim=(1, 2, 3))).view(ns, B).sum(dim=1)    # [ns]
    recon = output_sample[(ns - 1) * B:]
    if deterministic:
        kl = lp.new_zeros(B)
    elif use_normal:
        kl = sum(kl_normal_std(loc, scale).sum(dim=(1, 2, 3)) for loc, scale in q[1:])   # the input level is unused
    else:                                                           # prior Beta(0.5, 0.5), ctvae/main_ct_vae.py:372
        kl = sum(torch.distributions.kl_divergence(d, torch.distributions.Beta(torch.full_like(d.concentration1, 0.5),
                                                                             torch.full_like(d.concentration1, 0.5)))
                 .sum(dim=(1, 2, 3)) for d in q[1:])
    loglik = log_prob_M.mean(dim=0)                                                       # scalar
    return kl_anneal * kl_multiplier * kl - loglik, kl, loglik, recon


# ---------------------------------------------------------------------------------------------------------
# driver
# ---------------------------------------------------------------------------------------------------------
class AngleStream:
    """Shuffled, repeating stream of 0..A-1 in chunks of `api` (ctvae/helper_functions.py:104-107)."""

    def __init__(self, num_angles, api, seed):
        self.n, self.api, self.rng, self.buf = num_angles, api, np.random.default_rng(seed), np.empty(0, np.int64)

    def next(self):
        while self.buf.size < self.api:
            self.buf = np.concatenate([self.buf, self.rng.permutation(self.n)])
        out, self.buf = self.buf[:self.api], self.buf[self.api:]
        return out


class PVAETrainer:
    def __init__(self, args, device):
        self.args, self.dev = args, device
        self.world, self.rank, _ = sharding.env_world()
        self.sqrt_reg = EPS32
        # the nets' shapes are static: --miopen_find lets MIOpen search its convolution algorithms once (14.9 -> 11.5 ms
        # per step at batch 5 on the MI355X, after a search that costs tens of seconds; channels-last was measured too
        # and loses)
        if getattr(args, "miopen_find", False):
            torch.backends.cudnn.benchmark = True
        torch.manual_seed(1234 + self.rank)
        a = args
        self.pnm_anneal = math.exp(math.log(a.pnm / a.pnm_start) / max(a.num_iter, 1)) if a.pnm_start else 1.0
        self._make_data()
        fm = [int(a.nfm * a.nfmm ** i) for i in range(a.num_blocks)]
        fmm = 1 if a.deterministic else 2
        self.enc = EncodeNet(len(a.algorithms) + 1, fm, fmm, a.kernel_size, a.stride_encode, a.il, a.ik).to(device)
        self.dec = DecodeNet(self.enc.channels, fmm, 1, a.kernel_size, a.stride_encode, a.il, a.ik).to(device)
        if self.world > 1:   # identical initial weights on every rank
            for p in list(self.enc.parameters()) + list(self.dec.parameters()):
                torch.distributed.broadcast(p.data, 0)
        self.params = list(self.enc.parameters()) + list(self.dec.parameters())
        self._bucket = None   # the gradients' resident flat bucket (built on the first step)
        self.pnm = torch.tensor(float(a.pnm), device=device, requires_grad=bool(a.train_pnm))
        # fused: the whole Adam update in one multi-tensor launch on the device (the default foreach form is ~10)
        self.opt = torch.optim.Adam(self.params + ([self.pnm] if a.train_pnm else []), lr=a.lr, eps=a.adam_epsilon,
                                    fused=(device.type == "cuda") and os.environ.get("CTPVAE_ADAM_FUSED", "1") == "1")
        self.kl_anneal = 1.0
        self.angles = AngleStream(self.num_angles, a.api, seed=7)     # same stream on every rank
        self.iter = 0

    # -- dataset: --input_path (the reference's dataset_<name>/ folder) or a synthetic foam set -----------------
    def _make_data(self):
        a, dev = self.args, self.dev
        self.pad = not a.no_pad
        if a.input_path:
            # x_train_sinograms.npy + dataset_parameters.npy as scripts/images_to_sinograms.py writes them
            # (ctvae/main_ct_vae.py:152-161); the phantoms themselves are not part of that folder
            sino_np, self.theta_np, self.P = dataset_io.get_sinograms(a.input_path)
            a.td = min(a.td, sino_np.shape[0])
            sino = torch.from_numpy(np.ascontiguousarray(sino_np[:a.td], dtype=np.float32)).to(dev)
            self.theta_np = np.asarray(self.theta_np, dtype=np.float64)
            self.num_angles = a.num_angles = len(self.theta_np)
            imgs = None
        else:
            N = a.n_pixel
            self.theta_np = phantoms.dense_theta(a.num_angles)
            self.num_angles = a.num_angles
            self.P = num_proj_pix(N, N) if self.pad else N
            imgs = phantoms.foam_batch(a.td, N, seed=0, supersample=2)
            sino = create_sinograms(torch.from_numpy(imgs).to(dev), self.theta_np, pad=self.pad).clamp_min(0)   # [td][A][P]
        # ctvae/main_ct_vae.py:156-161: --no_pad sinograms are as wide as the object
        self.x_size = self.y_size = int(math.floor(self.P / math.sqrt(2) - 2)) if self.pad else self.P
        # dose masks and sparse noisy measurements, ctvae/create_masks.py:45-95 (simulated at the FINAL pnm)
        # with --save_path the three setup arrays are written there when training and read back when not (the
        # reference's train / restore split, ctvae/create_masks.py:70,101-103, ctvae/helper_functions.py:523-526)
        drawing = bool(a.train) or not a.save_path
        # every rank draws the same arrays (same seeds); only rank 0 writes them -- concurrent np.save of one path from
        # several ranks is a truncate-and-write race
        save_here = a.save_path if (self.rank == 0 or not drawing) else None
        masks, self.proj_samples = create_all_masks(sino, a.num_angles, save_path=save_here, poisson_noise_multiplier=a.pnm,
                                                    num_sparse_angles=a.nsa, random=a.random, train=drawing, real_data=a.real_data,
                                                    toy_masks=a.toy_masks,
                                                    truncate_dataset=a.td, device=dev)
        self.masks, self.truth = masks, (torch.from_numpy(imgs).to(dev) if imgs is not None else None)
        # initial reconstructions for the encoder, ctvae/helper_functions.py:477-529 with FBP on the GPU
        enc_in = iradon_all(self.proj_samples, masks, self.P, self.theta_np, list(a.algorithms), self.sqrt_reg, self.x_size,
                            self.y_size, save_path=save_here, train=drawing)          # [td][X][Y][len(algorithms) + 1]
        if self.world > 1 and drawing and a.save_path:
            torch.distributed.barrier()                                                # files complete before anyone goes on
        self.input_encode = enc_in.permute(0, 3, 1, 2).contiguous()                   # [td][2][X][Y]
        self.theta = torch.from_numpy(self.theta_np.astype(np.float32)).to(dev)
        # the projector gets the HOST angle list: its tables and gather plan are then built once, on the host's bits, and
        # every step only passes its angle subset as an index operand
        self.theta_host = np.ascontiguousarray(self.theta_np, dtype=np.float32)
        self.order = np.random.default_rng(11)

    def _batch(self):
        """Global batch of `-b` examples, the same on every rank; each rank keeps its shard."""
        idx = self.order.choice(self.args.td, size=self.args.batch_size, replace=False)
        lo, hi = sharding.shard_range(len(idx), self.rank, self.world)
        idx = self._to_device(idx[lo:hi])
        return self.proj_samples[idx], self.masks[idx], self.input_encode[idx]

    def _to_device(self, host_array):
        """Small per-step index arrays go up through pinned memory without waiting: a plain torch.as_tensor(...,
        device=) is a blocking copy that first drains the stream, i.e. idles the GPU while the host prepares a step."""
        t = torch.from_numpy(np.ascontiguousarray(host_array))
        if self.dev.type != "cuda":
            return t
        return t.pin_memory().to(self.dev, non_blocking=True)

    # -- one step (CT_VAE.train_step, ctvae/main_ct_vae.py:463-486) -------------------------------------------
    def _next_inputs(self):
        """Host side of a step: the batch, the angle subset and the two annealed scalars."""
        a = self.args
        proj_sample, mask, input_encode = self._batch()
        # the step's angle subset stays in HOST memory: the projector kernels carry it in their launch arguments (no upload,
        # nothing on the stream in front of the forward)
        angles_i = torch.from_numpy(np.ascontiguousarray(self.angles.next().astype(np.int32)))
        pnm_factor = self.pnm_anneal ** self.iter
        self.kl_anneal = min(max(self.kl_anneal * a.klaf, 0.0), 100.0)
        return proj_sample, mask, input_encode, angles_i, pnm_factor, self.kl_anneal

    def _loss_and_update(self, proj_sample, mask, input_encode, angles_i, pnm_factor, kl_anneal):
        """Device side of a step: ELBO, backward, NaN filter + clip, Adam."""
        a = self.args
        pnm_i = self.pnm * pnm_factor
        loss_vec, kl, loglik, _ = find_loss_vae_unsup(proj_sample, mask, input_encode, self.enc, self.dec, pnm_i,
                                                      self.sqrt_reg, kl_anneal, a.klm, num_samples=a.ns,
                                                      theta=self.theta_host, angles_i=angles_i, pad=self.pad,
                                                      deterministic=a.deterministic, use_normal=a.use_normal)
        # ctvae/main_ct_vae.py:478 reduce_mean(loss_M_VAE) / 1e5 = mean_b(KL term) - loglik, where loglik already sums
        # over the batch.  Written so that the ranks' losses ADD UP to the global one (gradients are summed over ranks):
        # each rank contributes its objects' KL / global_B and its own objects' log-likelihood.
        del loss_vec
        loss = ((kl_anneal * a.klm * kl).sum() / a.batch_size - loglik) / 1e5
        self.opt.zero_grad(set_to_none=True)
        # one calling-thread pass over the graph: the default engine hands every backward() to a per-device worker thread
        # (~50 us of hand-off, and the Python backward of the projector node then runs behind the GIL of that thread)
        with torch.autograd.set_multithreading_enabled(False):
            loss.backward()
        with_grad = [p for p in self.opt.param_groups[0]["params"] if p.grad is not None]
        grads = [p.grad for p in with_grad]
        # ONE resident flat bucket (round 4): a multi-tensor copy in, one all-reduce on that memory, the filter and the clip on its
        # slices, and p.grad pointed at the slices -- no torch.cat, no per-tensor copy back (sharding.FlatGradBucket)
        if self._bucket is None or not self._bucket.matches(grads):
            self._bucket = sharding.FlatGradBucket(grads)
        bucket = self._bucket.fill(grads).allreduce_(average=False)
        # tf.where(is_nan, 0, grad), then tf.clip_by_norm(g, norm) per tensor (ctvae/main_ct_vae.py:482-484) -- in one
        # flat buffer and a handful of multi-tensor launches, with no host round trip (a Python `if norm > clip` per
        # tensor would synchronise 76 times a step)
        torch.nan_to_num_(bucket.flat, nan=0.0)
        norms = torch.stack(torch._foreach_norm(bucket.views))
        scales = (a.norm / norms).clamp_(max=1.0)                          # norm / max(l2, norm); l2 = 0 -> 1
        torch._foreach_mul_(bucket.views, list(scales.unbind()))
        bucket.attach(with_grad)
        self.opt.step()
        return loss.detach()

    def train_step(self, sync=True):
        """One optimisation step.  sync=True returns the loss as a python float (a host round trip, as the reference's
        per-iteration `.numpy()`); sync=False returns it as a 0-d device tensor so that the host keeps queueing the
        next step's ~800 launches while the GPU finishes this one (`train` reads the losses back in blocks)."""
        loss = self._loss_and_update(*self._next_inputs())
        self.iter += 1
        if self.world > 1:
            loss = loss.clone()
            torch.distributed.all_reduce(loss)
        return float(loss.item()) if sync else loss

    def train(self):
        a = self.args
        losses, pending, t0 = [], [], time.time()
        every = max(a.num_iter // 10, 1)
        for it in range(a.num_iter):
            pending.append(self.train_step(sync=False))
            report = it % every == 0 or it == a.num_iter - 1
            saving = bool(a.save_path) and (it % a.si == 0 or it == a.num_iter - 1)
            if report or saving or len(pending) >= 32:
                # losses come back in blocks: one host round trip per block instead of one per iteration; the
                # reference's NaN stop (ctvae/main_ct_vae.py:401-402) therefore acts within 32 iterations
                losses += torch.stack(pending).tolist()
                pending = []
                if any(math.isnan(v) for v in losses[-32:]):
                    raise SystemExit("loss is NaN")
            if report and self.rank == 0:
                print(f"Iteration number: {it}  Training loss_M_VAE: {losses[-1]:.6f}", flush=True)
            if saving and self.rank == 0:
                self.save(os.path.join(a.save_path, "training_checkpoints", f"ckpt-{it}.pt"), losses)
                np.save(os.path.join(a.save_path, "train_loss_vec.npy"), np.asarray(losses))      # ctvae/main_ct_vae.py:413
        if a.save_path and self.rank == 0 and a.num_iter > 0:
            np.save(os.path.join(a.save_path, "training_time.npy"), (time.time() - t0) / 60)       # minutes, :421-422
        return losses, time.time() - t0

    def save(self, path, losses):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save({"enc": self.enc.state_dict(), "dec": self.dec.state_dict(), "opt": self.opt.state_dict(),
                    "kl_anneal": self.kl_anneal, "pnm": self.pnm.detach().cpu(), "iter": self.iter, "losses": losses}, path)

    def latest_checkpoint(self):
        """Path of the newest ckpt-<iteration>.pt under --save_path/training_checkpoints."""
        folder = os.path.join(self.args.save_path or ".", "training_checkpoints")
        names = [n for n in (os.listdir(folder) if os.path.isdir(folder) else []) if n.startswith("ckpt-") and n.endswith(".pt")]
        if not names:
            raise FileNotFoundError(f"no checkpoint under {folder}")
        return os.path.join(folder, max(names, key=lambda n: int(n[5:-3])))

    def restore(self, path):
        ck = torch.load(path, map_location=self.dev)
        self.enc.load_state_dict(ck["enc"])
        self.dec.load_state_dict(ck["dec"])
        self.opt.load_state_dict(ck["opt"])
        self.kl_anneal, self.iter = ck["kl_anneal"], ck["iter"]
        with torch.no_grad():
            self.pnm.copy_(ck["pnm"].to(self.dev))

    @torch.no_grad()
    def evaluate(self, n=None):
        """Mean decoder output on the first n examples vs. the phantoms (MSE), and the FBP input's MSE."""
        n = n or min(self.args.td, 16)
        skips = self.enc(self.input_encode[:n] / 300)
        if self.args.deterministic:
            lat = skips
        elif self.args.use_normal:
            lat = [s.chunk(2, dim=1)[0] for s in skips]                                   # the latents' means
        else:                                                                             # Beta(a, b): a / (a + b)
            ab = [(positive_range(s.chunk(2, dim=1)[0]), positive_range(s.chunk(2, dim=1)[1])) for s in skips]
            lat = [a_ / (a_ + b_) for a_, b_ in ab]
        alpha, beta = self.dec(lat)
        rec = positive_range(alpha)[:, 0]
        if not self.args.use_normal:
            rec = rec / (rec + positive_range(beta)[:, 0])
        if self.truth is None:                    # a dataset folder holds sinograms only
            return float("nan"), float("nan")
        return float(((rec - self.truth[:n]) ** 2).mean()), float(((self.input_encode[:n, 0] - self.truth[:n]) ** 2).mean())

    @torch.no_grad()
    def final_evaluation(self, save_path=None):
        """CT_VAE.final_evaluation (ctvae/main_ct_vae.py:427-460): the unshuffled dataset in batches of `-b`, ALL angles,
        no update; keeps every batch's loss and one sample of the output distribution per object, and writes
        loss_final.npy / reconstruction_final.npy ([n][X][Y][1]) -- the file bin/final_merit.py scores."""
        a = self.args
        # full batches of `-b` on every rank (the log-likelihood sums over the batch, so a sharded batch would change the
        # per-batch losses the reference reports); the evaluation runs once and only rank 0 writes
        nb = a.batch_size
        losses, recons = [], []
        for k in range(0, (a.td // nb) * nb, nb):
            sl = slice(k, k + nb)
            loss_vec, _, _, recon = find_loss_vae_unsup(self.proj_samples[sl], self.masks[sl], self.input_encode[sl], self.enc,
                                                        self.dec, self.pnm, self.sqrt_reg, self.kl_anneal, a.klm,
                                                        num_samples=a.ns, theta=self.theta_host, angles_i=None, pad=self.pad,
                                                        deterministic=a.deterministic, use_normal=a.use_normal)
            losses.append(loss_vec.mean() / 1e5)
            recons.append(recon.permute(0, 2, 3, 1))
        loss_final = torch.stack(losses).cpu().numpy()
        reconstruction_final = torch.cat(recons).cpu().numpy()
        if save_path is not None and self.rank == 0:
            os.makedirs(save_path, exist_ok=True)
            np.save(os.path.join(save_path, "loss_final.npy"), loss_final)
            np.save(os.path.join(save_path, "reconstruction_final.npy"), reconstruction_final)
        return loss_final, reconstruction_final


def get_args(argv=None):
    """The reference's flags that reach the path (ctvae/main_ct_vae.py:30-116), same spellings and defaults."""
    p = argparse.ArgumentParser(description="P-VAE training with the MI355X projector")
    p.add_argument("--ae", type=float, dest="adam_epsilon", default=1e-7)
    p.add_argument("-b", type=int, dest="batch_size", default=4)
    p.add_argument("--ns", type=int, dest="ns", default=2)
    p.add_argument("--det", action="store_true", dest="deterministic")
    p.add_argument("-i", type=int, dest="num_iter", default=100)
    p.add_argument("--ik", type=int, dest="ik", default=4)
    p.add_argument("--il", type=int, dest="il", default=2)
    p.add_argument("--klaf", type=float, dest="klaf", default=1.0)
    p.add_argument("--klm", type=float, dest="klm", default=1.0)
    p.add_argument("--ks", type=int, dest="kernel_size", default=4)
    p.add_argument("--lr", type=float, dest="lr", default=1e-4)
    p.add_argument("--nb", type=int, dest="num_blocks", default=3)
    p.add_argument("--nfm", type=int, dest="nfm", default=20)
    p.add_argument("--nfmm", type=float, dest="nfmm", default=1.1)
    p.add_argument("--norm", type=float, dest="norm", default=100.0)
    p.add_argument("--normal", action="store_true", dest="use_normal",
                   help="Normal latents and a TruncatedNormal output distribution (every README recipe); without it the "
                        "reference's default Beta latents / Beta(0.5, 0.5) prior / Beta output (ctvae/main_ct_vae.py:69, :369-372)")
    p.add_argument("--nsa", type=int, dest="nsa", default=10)
    p.add_argument("--api", type=int, dest="api", default=5)
    p.add_argument("--pnm", type=float, dest="pnm", default=(2 ** 16 - 1) * 0.41)
    p.add_argument("--pnm_start", type=float, dest="pnm_start", default=None)
    p.add_argument("--train_pnm", action="store_true")
    p.add_argument("--random", action="store_true")
    p.add_argument("--save_path", default=None)
    p.add_argument("--restore", action="store_true", help="restore the latest checkpoint under --save_path before training / evaluating")
    p.add_argument("--ulc", action="store_true", dest="use_latest_ckpt", help="accepted: --restore always takes the latest checkpoint")
    p.add_argument("--input_path", default=None,
                   help="dataset folder written by scripts/images_to_sinograms.py (x_train_sinograms.npy, "
                        "dataset_parameters.npy); without it a seeded synthetic foam set of --td phantoms is made")
    p.add_argument("--real", action="store_true", dest="real_data", help="real data: no simulated noise (ctvae/main_ct_vae.py:105)")
    p.add_argument("--no_pad", action="store_true", help="sinograms have no zero-padding (ctvae/main_ct_vae.py:107)")
    p.add_argument("--toy_masks", action="store_true", help="the toy problem's two-angle masks (ctvae/main_ct_vae.py:109)")
    p.add_argument("--no_final_eval", action="store_true", help="skip the final evaluation (ctvae/main_ct_vae.py:113)")
    p.add_argument("--se", type=int, dest="stride_encode", default=2)
    p.add_argument("--si", type=int, dest="si", default=100000)
    p.add_argument("--miopen_find", action="store_true",
                   help="search MIOpen's convolution algorithms once (torch.backends.cudnn.benchmark); not in the reference")
    p.add_argument("--td", type=int, dest="td", default=100)
    p.add_argument("--algorithms", nargs="+", default=["gridrec"],
                   help="initial reconstructions fed to the encoder, one channel each (ctvae/main_ct_vae.py:111-112, same "
                        "default); on the GPU: gridrec (csrc/gridrec.hip, tomopy's default parzen filter), sirt, fbp "
                        "(ct_pvae_amd/recon.py), tv (flagged stand-in)")
    p.add_argument("--train", action="store_true")
    # synthetic-data knobs (the reference reads these from its dataset folder)
    p.add_argument("--n_pixel", type=int, default=128)
    p.add_argument("--num_angles", type=int, default=180)
    return p.parse_args(argv)


def main(argv=None):
    args = get_args(argv)
    world, rank, local = sharding.init_from_env()
    dev = torch.device("cuda", local)
    tr = PVAETrainer(args, dev)
    if args.restore:                                   # ctvae/main_ct_vae.py:363-368 (--ulc: the latest checkpoint)
        tr.restore(tr.latest_checkpoint())
    losses, secs = tr.train() if args.train else ([], 0.0)
    if not args.no_final_eval:
        loss_final, _ = tr.final_evaluation(args.save_path)
        if rank == 0:
            print(f"Average loss final : {float(loss_final.mean()):.6f}")
    if rank == 0:
        mse, mse_fbp = tr.evaluate()
        if losses:
            print(f"{len(losses)} iterations in {secs:.1f} s ({len(losses) / secs:.2f} it/s); loss {losses[0]:.5f} -> {losses[-1]:.5f}")
        print(f"MSE reconstruction {mse:.5f} (FBP input {mse_fbp:.5f})")
    return losses


if __name__ == "__main__":
    main()
