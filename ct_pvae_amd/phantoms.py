"""Seeded synthetic inputs for tests and benchmarks (numpy, host side).

xdesign (used by scripts/create_foam_images.py:24-40 of the reference) is not installable offline, so this is a
build-owned stand-in with the same recipe: a unit disc of value 1 holding non-overlapping circular pores of value 0
with radii U[size_lower, size_upper] (in units of the image side), a per-image target porosity drawn from the
generator, and area-averaged (supersampled) edges.  Values are float32 in [0, 1]."""
import numpy as np

__all__ = ["foam_phantom", "foam_batch", "dense_theta", "sparse_angle_indices", "toy_images"]


def foam_phantom(n_pixel, rng, size_lower=0.01, size_upper=0.2, supersample=8, max_tries=400):
    porosity = rng.random()
    target = porosity * np.pi * 0.25
    circles, area, tries = [], 0.0, 0
    while area < target and tries < max_tries:
        tries += 1
        r = rng.uniform(size_lower, size_upper)
        x, y = rng.random(2)
        if np.hypot(x - 0.5, y - 0.5) + r > 0.5:
            continue
        if any(np.hypot(x - cx, y - cy) < r + cr for cx, cy, cr in circles):
            continue
        circles.append((x, y, r))
        area += np.pi * r * r
    m = n_pixel * supersample
    c = (np.arange(m, dtype=np.float64) + 0.5) / m
    xx, yy = np.meshgrid(c, c, indexing="ij")
    img = ((xx - 0.5) ** 2 + (yy - 0.5) ** 2 <= 0.25).astype(np.float32)
    for cx, cy, r in circles:
        i0, i1 = max(int((cx - r) * m) - 1, 0), min(int((cx + r) * m) + 2, m)
        j0, j1 = max(int((cy - r) * m) - 1, 0), min(int((cy + r) * m) + 2, m)
        sub = (xx[i0:i1, j0:j1] - cx) ** 2 + (yy[i0:i1, j0:j1] - cy) ** 2 <= r * r
        img[i0:i1, j0:j1][sub] = 0.0
    return img.reshape(n_pixel, supersample, n_pixel, supersample).mean(axis=(1, 3)).astype(np.float32)


def foam_batch(batch, n_pixel=128, seed=0, supersample=8):
    """[batch][n_pixel][n_pixel] float32; numpy.random.default_rng(seed)."""
    rng = np.random.default_rng(seed)
    return np.stack([foam_phantom(n_pixel, rng, supersample=supersample) for _ in range(batch)], axis=0)


def dense_theta(num_angles=180):
    """np.linspace(0, pi, num_angles, endpoint=False), scripts/images_to_sinograms.py:34."""
    return np.linspace(0, np.pi, num_angles, endpoint=False)


def sparse_angle_indices(num_angles=180, num_sparse_angles=20):
    """The uniform mask of ctvae/create_masks.py:55-59: start 0, spacing ceil(num_angles/num_sparse_angles)."""
    spacing = int(np.ceil(num_angles / num_sparse_angles))
    return (np.arange(0, spacing * num_sparse_angles, spacing) % num_angles).astype(np.int64)


def toy_images():
    """The two 2x2 images of scripts/create_toy_images.py:36-40."""
    return np.stack([np.array([[1, 2], [3, 4]]) / 10, np.array([[3, 4], [1, 2]]) / 10]).astype(np.float32)
