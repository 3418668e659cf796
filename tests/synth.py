"""Deterministic synthetic frames (SURVEY.md Appendix B generator, reproduced verbatim
as the measurement contract: same seed => same bytes => the SHA-256 digests recorded
there apply)."""
import numpy as np


def synth(w, h, cf, bits, seed, frames=1, word_bytes=2):
    rng = np.random.default_rng(seed)
    cw = w if cf == '444' else w // 2
    ch = h // 2 if cf == '420' else h
    out = []
    for f in range(frames):
        for (pw, ph, c) in [(w, h, 0), (cw, ch, 1), (cw, ch, 2)]:
            y, x = np.mgrid[0:ph, 0:pw]
            base = (0.5 + 0.35 * np.sin(2 * np.pi * (x / pw * 3 + c * 0.3 + f * 0.1))
                    * np.cos(2 * np.pi * (y / ph * 2))) * (2 ** bits - 1)
            v = np.clip(np.rint(base + rng.normal(0, (2 ** bits) * 0.01, size=(ph, pw))),
                        0, 2 ** bits - 1).astype(np.uint16)
            if word_bytes == 2:
                out.append((v << (16 - bits)).astype('>u2').tobytes())
            else:
                out.append((v << (8 - bits)).astype(np.uint8).tobytes())
    return b''.join(out)


def noise_frame(w, h, cf, bits, seed, word_bytes=2, full_scale=False):
    """Uniform-noise frame (worst case for coefficient growth / code lengths)."""
    rng = np.random.default_rng(seed)
    cw = w if cf == '444' else w // 2
    ch = h // 2 if cf == '420' else h
    out = []
    for (pw, ph) in [(w, h), (cw, ch), (cw, ch)]:
        if full_scale:
            v = (rng.integers(0, 2, size=(ph, pw)) * (2 ** bits - 1)).astype(np.uint16)
        else:
            v = rng.integers(0, 2 ** bits, size=(ph, pw)).astype(np.uint16)
        if word_bytes == 2:
            out.append((v << (16 - bits)).astype('>u2').tobytes())
        else:
            out.append((v << (8 - bits)).astype(np.uint8).tobytes())
    return b''.join(out)
