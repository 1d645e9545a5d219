"""How many decisions of the point kernels depend on FMA contraction of  d = dx*dx + dy*dy + dz*dz ?

The reference's CUDA kernels are compiled by nvcc with contraction on (its default): the squared distance is evaluated as
fma(dz, dz, fma(dy, dy, dx*dx)) there, while the oracle (oracle/point_ops.c, -ffp-contract=off) and the HIP kernels
(point_ops.hip, -ffp-contract=off) round every product and every sum.  The two forms differ by an ulp now and then, which
matters only at the strict comparisons: ball query `d2 < r2` (ball_query.cu:34-47) and FPS's running arg-max
(sampling.cu:86-167).  This script evaluates both forms in exact f32 semantics (numpy f32 for the plain form; the fused
form through f64, where a product of two f32 is exact) on the clouds of the golden fixtures and counts the decisions that
differ.   python tools/study/fma_contraction.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G = os.path.join(ROOT, "tests", "golden")

def d2_plain(a, b):
    dx, dy, dz = (a[..., 0] - b[..., 0]), (a[..., 1] - b[..., 1]), (a[..., 2] - b[..., 2])
    return ((dx * dx).astype(np.float32) + (dy * dy).astype(np.float32)).astype(np.float32) + (dz * dz).astype(np.float32)

def fma32(a, b, c):  # round(a*b + c) once: a*b exact in f64 (24 + 24 bits), the sum rounded to f64 then f32
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)

def d2_fused(a, b):
    dx, dy, dz = (a[..., 0] - b[..., 0]), (a[..., 1] - b[..., 1]), (a[..., 2] - b[..., 2])
    return fma32(dz, dz, fma32(dy, dy, (dx * dx).astype(np.float32)))

def fps(pts, m, d2):
    n = pts.shape[0]
    dist = np.full(n, 1e38, np.float32)
    idx = np.zeros(m, np.int64)
    last = 0
    for j in range(1, m):
        d = d2(pts, pts[last][None, :]).astype(np.float32)
        dist = np.minimum(dist, d)
        last = int(np.argmax(dist))   # first maximum, like the tree reduction's tie rule on distinct values
        idx[j] = last
    return idx

def ball(centers, pts, r, u, d2):
    r2 = np.float32(r) * np.float32(r)
    out = np.zeros((centers.shape[0], u), np.int64)
    flips = 0
    for i, c in enumerate(centers):
        d = d2(pts, c[None, :])
        hit = np.nonzero(d < r2)[0][:u]
        if hit.size:
            out[i, :hit.size] = hit
            out[i, hit.size:] = hit[0]
    return out

clouds = []
z = np.load(os.path.join(G, "sa_module.npz")); clouds.append(("sa_module cloud (unit scale)", z["coords"][0].T.astype(np.float32)))
z = np.load(os.path.join(G, "ldm_e2e.npz"))
for i in range(2):
    clouds.append((f"ldm_e2e cloud {i} (normalised, /0.05)", z["pc"][i].astype(np.float32)))
z = np.load(os.path.join(G, "c5_ldm_e2e.npz")); clouds.append(("c5 partial cloud (4096 pts, duplicates)", z["pc"][0].astype(np.float32)))
tot = dict(pairs=0, differ=0, bq=0, bq_flip=0, fps=0, fps_flip=0)
for name, pts in clouds:
    n = pts.shape[0]
    m = n // 2
    a, b = fps(pts, m, d2_plain), fps(pts, m, d2_fused)
    first = int(np.argmax(a != b)) if (a != b).any() else -1
    ctr = pts[a[: n // 8]]
    pl, fu = d2_plain(pts[None, :, :], ctr[:, None, :]), d2_fused(pts[None, :, :], ctr[:, None, :])
    ulp = int((pl != fu).sum())
    flips = {}
    for r in (0.2, 0.4, 1.0):
        r2 = np.float32(r) * np.float32(r)
        flips[r] = int(((pl < r2) != (fu < r2)).sum())
    print(f"{name}: N={n}; d2 differs by an ulp in {ulp} of {pl.size} pairs ({100 * ulp / pl.size:.1f} %); "
          f"ball-query membership flips: " + ", ".join(f"r={r}: {v}" for r, v in flips.items())
          + f"; FPS ({m} of {n}): " + ("identical index sequence" if first < 0 else f"sequences diverge at pick {first}"))
