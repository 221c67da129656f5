"""Deterministic synthetic PCM for tests and bench.

Per stream (seed = 0x484D5033 + stream index, PCG64): 12 sinusoids f ~ U(60, 9000) Hz,
amplitude ~ U(0.02, 0.15), each with 0.1-2 Hz AM; + 0.1 x high-passed random-walk noise;
R = rho * L + (1 - rho) * R'; optional decaying white bursts every 0.7 s (to provoke short
blocks); normalised to 0.95 full scale; int16, interleaved L/R.
"""
import numpy as np

SEED0 = 0x484D5033


def _channel(rng, n, sr):
    t = np.arange(n, dtype=np.float64) / sr
    x = np.zeros(n)
    f = rng.uniform(60.0, 9000.0, 12)
    a = rng.uniform(0.02, 0.15, 12)
    fm = rng.uniform(0.1, 2.0, 12)
    ph = rng.uniform(0.0, 2 * np.pi, 12)
    for k in range(12):
        x += a[k] * (0.6 + 0.4 * np.sin(2 * np.pi * fm[k] * t)) * np.sin(2 * np.pi * f[k] * t + ph[k])
    w = np.cumsum(rng.standard_normal(n))
    w = w - np.convolve(w, np.ones(64) / 64.0, mode="same")  # high-pass the random walk
    w /= (np.abs(w).max() + 1e-9)
    return x + 0.1 * w


def stream_pcm(stream, nframes, sr=44100, rho=0.7, bursts=False):
    """int16 array [nframes*1152, 2]"""
    rng = np.random.Generator(np.random.PCG64(SEED0 + int(stream)))
    n = nframes * 1152
    left = _channel(rng, n, sr)
    rp = _channel(rng, n, sr)
    if bursts:
        period = int(0.7 * sr)
        env = np.exp(-np.arange(2000) / 200.0)
        for s0 in range(period // 2, n - 2000, period):
            left[s0:s0 + 2000] += 0.5 * env * rng.standard_normal(2000)
            rp[s0:s0 + 2000] += 0.5 * env * rng.standard_normal(2000)
    right = rho * left + (1.0 - rho) * rp
    x = np.stack([left, right], axis=1)
    x *= 0.95 / np.abs(x).max()
    return np.round(x * 32767.0).astype(np.int16)


def batch_pcm(nstreams, nframes, sr=44100, rho=0.7, bursts=False, first=0, unique=None):
    """int16 array [nstreams, nframes*1152, 2].  `unique` limits the number of distinct
    generated streams (the rest are time-rotated copies) to keep host-side generation cheap."""
    unique = nstreams if unique is None else min(unique, nstreams)
    base = [stream_pcm(first + i, nframes, sr, rho, bursts) for i in range(unique)]
    out = np.empty((nstreams, nframes * 1152, 2), dtype=np.int16)
    for i in range(nstreams):
        b = base[i % unique]
        out[i] = b if i < unique else np.roll(b, 1152 * 7 * (i // unique), axis=0)
    return out
