#!/usr/bin/env python3
"""CPU analysis (numpy, no GPU): how large are the pixel footprints of C4's chunks (10 M points 0.4*N(0,I) -> 512^2,
cloud cell-sorted into 16^3 Hilbert-numbered cells as dpr_coarse.h does inside the call), and would cutting the
4096-point chunks of the 2-D chunk-owner path (dpr_chunkown.hip) into smaller ones empty k_co_splat_wide's work list?
Footprint of a chunk under a pose as co_footprint computes it: projected centre +- sum_j |R[d,j]| h_j of the chunk's
bounding box, clipped to the image; "wide" = more than kCOCap = 9984 cells.  Result: profiles/r06_experiments.md section 6."""
import numpy as np
rng = np.random.default_rng(0)
P = 10_000_000
pts = (0.4 * rng.standard_normal(size=(P, 3), dtype=np.float32))
BITS = 4
def hilbert_key(X):  # Skilling: axes -> transpose; X: (n,3) uint32
    X = X.copy(); n = 3
    M = 1 << (BITS - 1)
    Q = M
    while Q > 1:
        Pm = Q - 1
        for i in range(n):
            sel = (X[:, i] & Q) != 0
            X[sel, 0] ^= Pm
            t = (X[:, 0] ^ X[:, i]) & Pm
            t[sel] = 0
            X[:, 0] ^= t; X[:, i] ^= t
        Q >>= 1
    for i in range(1, n): X[:, i] ^= X[:, i - 1]
    t = np.zeros(len(X), dtype=np.uint32)
    Q = M
    while Q > 1:
        sel = (X[:, n - 1] & Q) != 0
        t[sel] ^= Q - 1
        Q >>= 1
    for i in range(n): X[:, i] ^= t
    key = np.zeros(len(X), dtype=np.uint32)
    for b in range(BITS - 1, -1, -1):
        for i in range(n):
            key = (key << 1) | ((X[:, i] >> b) & 1)
    return key
x = (pts * 0.5 + 0.5) * 16
Xc = np.clip(np.floor(x), 0, 15).astype(np.uint32)
key = hilbert_key(Xc)
order = np.argsort(key, kind='stable')
sp = pts[order]
nch = (P + 4095) // 4096
rngp = np.random.default_rng(1)
def boxes(sz):
    n = P // sz
    a = sp[:n * sz].reshape(n, sz, 3)
    lo = a.min(1); hi = a.max(1)
    return 0.5 * (lo + hi), 0.5 * (hi - lo)
res = {}
Rs = [np.linalg.qr(rngp.standard_normal((3, 3)))[0] for _ in range(16)]
for sz in (4096, 2048, 1024, 512, 256):
    c, h = boxes(sz)
    wide = 0; tot = 0; cells_all = []
    for R in Rs:
        Rp = R[:2, :]  # projection rows
        pc = c @ Rp.T; ph = h @ np.abs(Rp).T
        a = (pc - ph + 1) * 256 - 2.5; b = (pc + ph + 1) * 256 + 1.5
        l = np.clip(np.floor(a), 0, 511); u = np.clip(np.floor(b) + 1, 0, 511)
        inside = (b > 0) & (a < 512)
        W = np.where(inside, u - l + 1, 0)
        cells = W[:, 0] * W[:, 1]
        cells_all.append(cells)
        wide += (cells > 9984).sum(); tot += (cells > 0).sum()
    cells_all = np.concatenate(cells_all)
    print(sz, "chunks", len(c), "wide frac of pairs %.4f" % (wide / max(tot, 1)), "empty frac %.3f" % (1 - tot / (len(c) * len(Rs))),
          "median cells", np.median(cells_all[cells_all > 0]), "p90", np.percentile(cells_all[cells_all > 0], 90))
    res[sz] = (c, h)
# for wide 4096-chunks: at which split level do they fit?
c, h = res[4096]
R = Rs[0]; Rp = R[:2]
def cells_of(c, h, Rp):
    pc = c @ Rp.T; ph = h @ np.abs(Rp).T
    a = (pc - ph + 1) * 256 - 2.5; b = (pc + ph + 1) * 256 + 1.5
    l = np.clip(np.floor(a), 0, 511); u = np.clip(np.floor(b) + 1, 0, 511)
    inside = ((b > 0) & (a < 512)).all(1)
    W = u - l + 1
    return np.where(inside, W[:, 0] * W[:, 1], 0)
wide_idx = np.where(cells_of(c, h, Rp) > 9984)[0]
print("pose0: wide chunks", len(wide_idx), "of", len(c))
# points in wide chunks
print("points in wide chunks frac", len(wide_idx) * 4096 / P)
for sz in (2048, 1024, 512, 256):
    cs, hs = res[sz]
    k = 4096 // sz
    sub = (wide_idx[:, None] * k + np.arange(k)[None, :]).ravel()
    sub = sub[sub < len(cs)]
    cc = cells_of(cs[sub], hs[sub], Rp)
    print(sz, "sub-chunks of wide chunks:", len(sub), "still wide frac %.3f" % ((cc > 9984).mean()), "empty %.3f" % ((cc == 0).mean()))
# exact projected bounds for the 4096 chunks, pose 0
n = P // 4096
a = sp[:n * 4096].reshape(n, 4096, 3) @ Rp.T
lo = a.min(1); hi = a.max(1)
aa = (lo + 1) * 256 - 2.5; bb = (hi + 1) * 256 + 1.5
l = np.clip(np.floor(aa), 0, 511); u = np.clip(np.floor(bb) + 1, 0, 511)
inside = ((bb > 0) & (aa < 512)).all(1)
W = u - l + 1
ce = np.where(inside, W[:, 0] * W[:, 1], 0)
print("exact projected bounds: wide frac %.4f" % ((ce > 9984).sum() / (ce > 0).sum()), "median", np.median(ce[ce > 0]))
