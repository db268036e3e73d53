"""Procedural stand-in for a real image set (there is no data set in the build container, and the reference's own data loader
reads an images.npy cache, train_rgbd.py:172-189): used by the long-run scripts and tests that need a discriminator with
something to model.  Uniform noise -- what the benchmark feeds the step, where only the arithmetic matters -- is a game D wins
outright and says nothing about training behaviour."""


def procedural_images(n, side, seed=0):
    """(n,3,side,side) uint8: one shaded ellipsoid per image on a vertical two-colour gradient."""
    import numpy as np
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:side, 0:side].astype("float32") / (side - 1) * 2 - 1
    out = np.empty((n, 3, side, side), dtype="uint8")
    for i in range(n):
        top, bot = rng.uniform(0.2, 1.0, 3), rng.uniform(0.0, 0.6, 3)
        t = ((yy + 1) / 2)[None]
        img = top[:, None, None] * (1 - t) + bot[:, None, None] * t
        cx, cy = rng.uniform(-0.3, 0.3, 2)
        a, b = rng.uniform(0.35, 0.7), rng.uniform(0.2, 0.45)
        th = rng.uniform(-0.5, 0.5)
        xr = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        yr = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        r2 = (xr / a) ** 2 + (yr / b) ** 2
        inside = r2 < 1
        shade = np.sqrt(np.clip(1 - r2, 0, 1)) * 0.8 + 0.2             # a lit ellipsoid: brightness ~ surface height
        col = rng.uniform(0.1, 1.0, 3)
        img = np.where(inside[None], col[:, None, None] * shade[None], img)
        out[i] = np.clip(img * 255 + rng.normal(0, 2.0, img.shape), 0, 255).astype("uint8")
    return out
