"""numpy mirror of the HIP path's counter-based dropout masks (csrc/common.h: drop_keep) -- TEST INFRASTRUCTURE ONLY.
Lets the CPU oracles apply exactly the masks the kernels regenerate, so training steps WITH dropout can be compared."""
import numpy as np
import torch


def keep_mask(seed, site, idx, p):
    """idx: integer array of element indices -> bool array (True = kept)"""
    p32 = np.float32(p)
    t = float(p32) * 4294967296.0
    thresh = np.uint32(4294967295 if t >= 4294967295.0 else int(t))
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32) * np.uint32(0x9E3779B1)
        x ^= (idx >> np.uint64(32)).astype(np.uint32) * np.uint32(0x85EBCA77)
        x ^= np.uint32((int(seed) * 0xC2B2AE3D) & 0xFFFFFFFF)
        x ^= np.uint32((int(site) * 0x27D4EB2F) & 0xFFFFFFFF)
        x ^= x >> np.uint32(16); x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15); x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return x >= thresh


def scale_of(p):
    return float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))


class Dropper(object):
    """Hooks for the oracles: ``rows`` = elementwise dropout of a (rows, n) activation whose first row is global row ``row0``;
    ``att`` = dropout of attention probabilities (n_head, Lq, Lk) of a sequence whose first query is global row ``row0``."""

    def __init__(self, p, seed):
        self.p, self.seed = p, seed

    def rows(self, site, x, row0):
        r, n = x.shape
        idx = (np.arange(r, dtype=np.uint64)[:, None] + np.uint64(row0)) * np.uint64(n) + np.arange(n, dtype=np.uint64)[None, :]
        m = torch.from_numpy(keep_mask(self.seed, site, idx, self.p)).to(x.dtype) * scale_of(self.p)
        return x * m

    def att(self, site, probs, row0):
        h, lq, lk = probs.shape
        q = np.arange(lq, dtype=np.uint64)[None, :, None] + np.uint64(row0)
        hh = np.arange(h, dtype=np.uint64)[:, None, None]
        k = np.arange(lk, dtype=np.uint64)[None, None, :]
        idx = ((q * np.uint64(h) + hh) << np.uint64(12)) + k
        m = torch.from_numpy(keep_mask(self.seed, site, idx, self.p)).to(probs.dtype) * scale_of(self.p)
        return probs * m
