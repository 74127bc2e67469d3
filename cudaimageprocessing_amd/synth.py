"""Deterministic synthetic images: counter-based, so any shard of any frame can
be generated independently on any rank, on the host (numpy, here) or on the device
(ops.synth -> mi_synth_u8x4_dev), and is identical everywhere."""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def synth_pixels(seed, first_idx, count):
    """uint32 pixels [first_idx, first_idx+count) of stream `seed` (RGBA little-endian, A=255)."""
    with np.errstate(over="ignore"):
        idx = np.arange(first_idx, first_idx + count, dtype=np.uint64)
        z = np.uint64(seed) * _GOLD + idx + _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z ^= z >> np.uint64(31)
    return (z & np.uint64(0xFFFFFFFF)).astype(np.uint32) | np.uint32(0xFF000000)


def synth_image(seed, w, h, first_idx=0):
    return synth_pixels(seed, first_idx, w * h).view(np.uint8).reshape(h, w, 4)


def pattern_image(kind, w, h, seed=0):
    """Structured test patterns: 'constant', 'gradient', 'checker', 'impulse', 'binary', 'natural'."""
    y, x = np.mgrid[0:h, 0:w]
    img = np.empty((h, w, 4), np.uint8)
    if kind == "constant":
        img[..., :3] = (seed * 37 + 101) % 256
    elif kind == "gradient":
        img[..., 0] = (x * 255 // max(1, w - 1))
        img[..., 1] = (y * 255 // max(1, h - 1))
        img[..., 2] = ((x + y) * 255 // max(1, w + h - 2))
    elif kind == "checker":
        c = (((x // 3) + (y // 2)) % 2 * 255).astype(np.uint8)
        img[..., 0] = c
        img[..., 1] = 255 - c
        img[..., 2] = c
    elif kind == "impulse":
        img[..., :3] = 0
        img[h // 2, w // 2, :3] = 255
        img[min(h - 1, 1), min(w - 1, 1), :3] = 200
    elif kind == "binary":
        r = synth_image(seed, w, h)
        img[..., :3] = np.where(r[..., :3] > 127, 255, 0)
    elif kind == "natural":
        r = synth_image(seed, w, h).astype(np.float32)
        base = 128 + 60 * np.sin(x / 17.0) + 50 * np.cos(y / 11.0)
        for c in range(3):
            img[..., c] = np.clip(base + (r[..., c] - 128) * 0.2 + 10 * c, 0, 255).astype(np.uint8)
    else:
        raise ValueError(kind)
    img[..., 3] = 255
    return img
