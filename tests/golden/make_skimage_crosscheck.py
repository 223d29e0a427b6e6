"""Generates tests/golden/skimage_crosscheck.npz -- an INDEPENDENT cross-check of the ray-driven projector (a7), the FBP
(a6) and gridrec (f3) restatements against scikit-image 0.18.3's radon / iradon, the a6/a7 analogue of
gridsample_crosscheck.npz.  scikit-image is not a parity target (another discretisation: rotate-with-interpolation and
column sums; its own centre conventions) -- it is a second opinion, written by other people, on orientation, angle sense,
centre and scale.

Run in the BUILD container only, under the leftover conda interpreter (the system python has no scikit-image):

    /opt/conda/bin/python3.9 tests/golden/make_skimage_crosscheck.py

Only the .npz (numeric arrays) is committed."""
import os

import numpy as np
from skimage.transform import iradon, radon
from skimage.transform.radon_transform import _get_fourier_filter

N = 128
yy, xx = np.mgrid[0:N, 0:N].astype(np.float64)
img = np.exp(-((yy - N * 0.4) ** 2 + (xx - N * 0.55) ** 2) / (2 * (N / 10) ** 2))
img += 0.5 * np.exp(-((yy - N * 0.65) ** 2 + (xx - N * 0.35) ** 2) / (2 * (N / 14) ** 2))
img[(yy - 90) ** 2 + (xx - 80) ** 2 < 49] += 0.7                     # a sharp asymmetric feature
theta_deg = np.arange(180.0)
sino = radon(img, theta=theta_deg, circle=False)                     # [182 bins][180 angles]
rec = iradon(sino, theta=theta_deg, filter_name="ramp", circle=False, output_size=N)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "skimage_crosscheck.npz")
np.savez_compressed(out, img=img.astype(np.float32), theta_deg=theta_deg, sk_sino=sino.astype(np.float32),
                    sk_rec=rec.astype(np.float32), ramp184=_get_fourier_filter(184, "ramp")[:, 0].astype(np.float64))
print("wrote", out, sino.shape, float(np.linalg.norm(rec - img) / np.linalg.norm(img)))
