"""Counter-based splitmix64 and the synthetic inputs of SURVEY.md §8(d).

Test infrastructure (see ``oracle/__init__.py``).  The same generator is
implemented on the device (``eioku_amd/csrc/synth.hip``) so that full-size bench
inputs never cross PCIe; ``tests/test_synth.py`` checks the two agree bit for bit.

Element ``i`` of stream ``seed`` is ``mix(seed + (i+1)*GOLDEN)`` - a pure function
of ``(seed, i)`` - so any slice can be generated independently.
"""

from __future__ import annotations

import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """``n`` 64-bit outputs of stream ``seed`` starting at element ``offset``."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def uniform_bytes(seed: int, n: int) -> np.ndarray:
    """``n`` uint8: byte ``j`` is byte ``j%8`` (little endian) of output ``j//8``."""
    words = splitmix64(seed, (n + 7) // 8)
    return words.view(np.uint8)[:n].copy()


def uniform_f32(seed: int, n: int) -> np.ndarray:
    """U[0,1) float32 with 24 random bits: ``(x >> 40) * 2**-24`` (exact)."""
    return (splitmix64(seed, n) >> np.uint64(40)).astype(np.float32) * np.float32(2.0**-24)


def approx_normal_f32(seed: int, n: int) -> np.ndarray:
    """Irwin-Hall(4) approximation of N(0,1) built from integer arithmetic only.

    The four 16-bit fields of one splitmix64 output are summed (an exact integer
    in [0, 4*65535]), centred and scaled by a float32 constant: bit-reproducible
    in numpy and on the device (one int->float conversion and one multiply).
    """
    x = splitmix64(seed, n)
    m = np.uint64(0xFFFF)
    s = (x & m) + ((x >> np.uint64(16)) & m) + ((x >> np.uint64(32)) & m) + (x >> np.uint64(48))
    c = s.astype(np.int64) - 131070  # centre: 4*65535/2
    # var of one field = (65536^2-1)/12 ; four fields -> sigma = 37837.2...
    return c.astype(np.float32) * np.float32(1.0 / 37837.22)


# ---------------------------------------------------------------------------
# Synthetic video (SURVEY.md §8d "Frames")
# ---------------------------------------------------------------------------

def scene_schedule(seed: int, n_frames: int, mean_len: int = 150, jitter: int = 60) -> np.ndarray:
    """Scene id per frame: scene lengths are ``mean_len + U{0..jitter-1}`` frames."""
    ids = np.empty(n_frames, dtype=np.int32)
    r = splitmix64(seed ^ 0x5CE2E, n_frames + 1)  # at most n_frames scenes
    t, s = 0, 0
    while t < n_frames:
        ln = mean_len + int(r[s] % np.uint64(max(jitter, 1)))
        ids[t:t + ln] = s
        t += ln
        s += 1
    return ids


def scene_params(seed: int, scene_ids: np.ndarray) -> np.ndarray:
    """Per-frame ``[base_b, base_g, base_r, gx, gy]`` int32 table, shape (n,5).

    base in [32,223], gx/gy in [0,31]; derived from the scene id only.
    """
    ns = int(scene_ids.max()) + 1 if scene_ids.size else 0
    r = splitmix64(seed ^ 0xBA5E, ns)
    tab = np.empty((ns, 5), dtype=np.int32)
    tab[:, 0] = 32 + ((r >> np.uint64(0)) & np.uint64(0xFF)) % np.uint64(192)
    tab[:, 1] = 32 + ((r >> np.uint64(8)) & np.uint64(0xFF)) % np.uint64(192)
    tab[:, 2] = 32 + ((r >> np.uint64(16)) & np.uint64(0xFF)) % np.uint64(192)
    tab[:, 3] = (r >> np.uint64(24)) & np.uint64(31)
    tab[:, 4] = (r >> np.uint64(32)) & np.uint64(31)
    return tab[scene_ids]


def synth_frames_bgr(seed: int, n: int, h: int, w: int, params: np.ndarray | None = None,
                     first_frame: int = 0) -> np.ndarray:
    """uint8 BGR frames (n,h,w,3): ``clip(base_c + ((x*gx + y*gy)>>10) + noise, 0, 255)``.

    noise = (byte % 9) - 4 where ``byte`` is element ``((t*h + y)*w + x)*3 + c`` of
    ``uniform_bytes(seed)`` (t is the absolute frame number ``first_frame + i``).
    ``params`` is the (n,5) table from :func:`scene_params`; default = one scene change
    schedule derived from ``seed``.
    """
    if params is None:
        # the schedule is a function of absolute frame numbers, so slices agree with the whole
        params = scene_params(seed, scene_schedule(seed, first_frame + n))[first_frame:]
    per = h * w * 3
    out = np.empty((n, h, w, 3), dtype=np.uint8)
    yy = np.arange(h, dtype=np.int32)[:, None]
    xx = np.arange(w, dtype=np.int32)[None, :]
    for i in range(n):
        t = first_frame + i
        # bytes [t*per, (t+1)*per) of the stream; per need not be a multiple of 8
        lo = t * per
        w0, w1 = lo // 8, (lo + per + 7) // 8
        words = splitmix64(seed, w1 - w0, offset=w0)
        b = words.view(np.uint8)[lo - w0 * 8: lo - w0 * 8 + per].reshape(h, w, 3)
        noise = (b % 9).astype(np.int32) - 4
        bb, bg, br, gx, gy = (int(v) for v in params[i])
        grad = (xx * gx + yy * gy) >> 10
        base = np.array([bb, bg, br], dtype=np.int32)[None, None, :]
        out[i] = np.clip(base + grad[:, :, None] + noise, 0, 255).astype(np.uint8)
    return out
