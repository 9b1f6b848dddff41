#!/usr/bin/env python3
"""How many steps would a cost-driven (dynamic-programme) collapse into W-wide nodes take, W = 4 and 8?  (VERDICT r2 item 1c and the
eight-wide question of round 2, asked again on a better binary tree.)  Binary tree = the oracle's: binned SAH down to single triangles
(sah_top 1, tree order).  C(n, m) after Ylitie et al. 2017; leaves of <= LEAF triangles.  Counts node visits and triangle tests of
primary rays and of cosine-distributed bounce rays leaving the primary hits, children visited nearest first.  Pure Python.
Test infrastructure (uses the oracle): lives under tests/."""
import sys, time, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import orc
from raytracer3_amd import scenes
LEAFB = 0x80000000
mesh = scenes.atrium(1.0)
osc = orc.Scene(mesh, leaf_size=1, node_width=2, quantized=0, sah_top=int(sys.argv[1]) if len(sys.argv) > 1 else 1, tree_order=1)
nodes = osc.nodes(); tris = osc.tris().view(np.float32).reshape(-1, 12)
nf = nodes.view(np.float32).reshape(-1, 16); nu = nodes.view(np.uint32).reshape(-1, 16)
N = len(nf)
cb = np.zeros((N, 2, 6), np.float64)
cb[:, 0] = nf[:, 0:6]; cb[:, 1] = nf[:, 6:12]
ref = nu[:, 12:14].astype(np.int64)
def harea(b): e = b[..., 3:6] - b[..., 0:3]; return e[..., 0] * e[..., 1] + e[..., 1] * e[..., 2] + e[..., 2] * e[..., 0]
nb = np.concatenate([np.minimum(cb[:, 0, :3], cb[:, 1, :3]), np.maximum(cb[:, 0, 3:], cb[:, 1, 3:])], 1)
A = harea(nb); CA = harea(cb)
# pre-order / counts / first triangle
order = []; st = [0]
while st:
    i = st.pop(); order.append(i)
    for c in (1, 0):
        if not ref[i, c] & LEAFB: st.append(int(ref[i, c]))
cnt = np.zeros(N, np.int64); first = np.zeros(N, np.int64)
for i in reversed(order):
    cnt[i] = sum(1 if ref[i, c] & LEAFB else cnt[ref[i, c]] for c in (0, 1))
for i in order:
    f = first[i] if i else 0
    for c in (0, 1):
        r = ref[i, c]
        if r & LEAFB: f += 1
        else: first[r] = f; f += cnt[r]
def collapse(W, LEAF, c_node=1.0, c_tri=1.0):
    C = np.zeros((N, W - 1)); leaf1 = np.zeros(N, bool); K = np.zeros((N, W + 1), np.int8)  # K[n, j] = k of D(n, j); 0 = "use C(n, j-1)"
    for i in reversed(order):
        cc = []
        for c in (0, 1):
            r = ref[i, c]
            cc.append(np.full(W - 1, CA[i, c] * c_tri) if r & LEAFB else C[r])
        Cl, Cr = cc
        D = {}
        for j in range(2, W + 1):
            best = None
            for k in range(1, j):
                if k > W - 1 or j - k > W - 1: continue
                v = Cl[k - 1] + Cr[j - k - 1]
                if best is None or v < best[0]: best = (v, k)
            D[j] = best
        cint = A[i] * c_node + D[W][0]; K[i, W] = D[W][1]
        cleaf = A[i] * cnt[i] * c_tri if (cnt[i] <= LEAF and i != 0) else np.inf
        leaf1[i] = cleaf <= cint
        C[i, 0] = min(cleaf, cint)
        for j in range(2, W):
            if D[j][0] < C[i, j - 2]: C[i, j - 1] = D[j][0]; K[i, j] = D[j][1]
            else: C[i, j - 1] = C[i, j - 2]; K[i, j] = 0
    wide = {}
    def expand(r, m, out):
        if r & LEAFB: out.append((None, int(r))); return
        while m > 1 and K[r, m] == 0: m -= 1
        if m == 1:
            out.append((None, (LEAFB | ((cnt[r] - 1) << 28) | first[r]) if leaf1[r] else int(r))); return
        k = K[r, m]
        expand(ref[r, 0], k, out); expand(ref[r, 1], m - k, out)
    st = [0]
    while st:
        i = st.pop(); out = []
        k = K[i, W]
        expand(ref[i, 0], k, out); expand(ref[i, 1], W - k, out)
        wide[i] = out
        for _, r in out:
            if not r & LEAFB: st.append(r)
    return wide, C[0, 0]
# boxes of slots: need box for each slot ref: a binary node's box = nb, a single leaf's = child box of its parent
leafbox = {}
for i in range(N):
    for c in (0, 1):
        if ref[i, c] & LEAFB: leafbox[int(ref[i, c]) & 0x0FFFFFFF] = cb[i, c]
def finish(wide):
    out = {}
    for i, sl in wide.items():
        bx = []; rf = []
        for _, r in sl:
            if r & LEAFB:
                f = r & 0x0FFFFFFF; c = ((r >> 28) & 7) + 1
                b = np.concatenate([np.min([leafbox[f + k][:3] for k in range(c)], 0), np.max([leafbox[f + k][3:] for k in range(c)], 0)])
            else: b = nb[r]
            bx.append(b); rf.append(r)
        out[i] = (np.array(bx), rf)
    return out
def tri_hit(o, d, f, c, tmin, best):
    for k in range(f, f + c):
        v0 = tris[k, 0:3].astype(np.float64); e1 = tris[k, 3:6] - v0; e2 = tris[k, 6:9] - v0
        pv = np.cross(d, e2); det = e1 @ pv
        if det == 0: continue
        inv = 1 / det; tv = o - v0; u = (tv @ pv) * inv
        if u < 0 or u > 1: continue
        qv = np.cross(tv, e1); v = (d @ qv) * inv
        if v < 0 or u + v > 1: continue
        t = (e2 @ qv) * inv
        if tmin < t < best: best = t
    return best, c
def traverse(wide, rays, any_hit=False):
    tn_ = tt_ = 0
    for j in range(rays.shape[1]):
        o = rays[0:3, j].astype(np.float64); d = rays[3:6, j].astype(np.float64); tmin = rays[6, j]; best = float(rays[7, j]); t0_ = best
        inv = 1 / np.where(np.abs(d) < 1e-20, 1e-20, d); st = [0]
        while st:
            r = st.pop()
            if r & LEAFB:
                best, c = tri_hit(o, d, r & 0x0FFFFFFF, ((r >> 28) & 7) + 1, tmin, best); tt_ += c
                if any_hit and best < t0_: break
                continue
            bx, rf = wide[r]; tn_ += 1
            a = (bx[:, 0:3] - o) * inv; b = (bx[:, 3:6] - o) * inv
            tn = np.maximum(np.minimum(a, b).max(1), tmin); tf = np.minimum(np.maximum(a, b).min(1), best)
            idx = np.nonzero(tn <= tf)[0]
            if len(idx) == 0: continue
            od = idx[np.argsort(-tn[idx] if any_hit else tn[idx], kind='stable')]
            for k in reversed(od): st.append(rf[k])
    return tn_ / rays.shape[1], tt_ / rays.shape[1]
W_, H_ = 1920, 1080
g = orc.camera_gconst(width=W_, height=H_, **scenes.ATRIUM_CAMERA)
rng = np.random.default_rng(1); NR = 1200
xs = rng.integers(0, W_, NR); ys = rng.integers(0, H_, NR)
pr = orc.primary_rays(g, xs, ys)
t, u, v, p = osc.trace_closest(pr)
hit = p != 0xFFFFFFFF
P = (pr[0:3] + pr[3:6] * t)[:, hit]; D0 = pr[3:6][:, hit]
# geometric normals of the hit triangles (tris are in tree order: find by prim id)
prim_of = tris[:, 9].view(np.uint32); pos = np.zeros(prim_of.max() + 1, np.int64); pos[prim_of] = np.arange(len(prim_of))
tp = tris[pos[p[hit]]]
nrm = np.cross(tp[:, 3:6] - tp[:, 0:3], tp[:, 6:9] - tp[:, 0:3]); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
nrm = nrm.T; fl = (nrm * D0).sum(0) > 0; nrm[:, fl] *= -1
u1 = rng.random(P.shape[1]); u2 = rng.random(P.shape[1]); ph = 2 * np.pi * u1; ct = np.sqrt(1 - u2); stt = np.sqrt(u2)
a = np.where(np.abs(nrm[0]) > 0.9, 0, 1); e = np.stack([a, 1 - a, np.zeros_like(a)]).astype(float)
b1 = np.cross(nrm.T, e.T); b1 /= np.linalg.norm(b1, axis=1, keepdims=True); b2 = np.cross(nrm.T, b1)
dd = (b1.T * np.cos(ph) * stt + b2.T * np.sin(ph) * stt + nrm * ct)
sec = np.concatenate([P, dd, np.full((1, P.shape[1]), 1e-3), np.full((1, P.shape[1]), 1e5)]).astype(np.float32)
print('rays', sec.shape[1])
for W, LEAF in ((4, 2), (8, 2), (8, 4), (16, 4)):
    t0 = time.time(); w, c = collapse(W, LEAF); wf = finish(w)
    fill = np.mean([len(v[1]) for v in wf.values()])
    print(f'W={W} leaf<={LEAF}: nodes {len(wf)} fill {fill:.2f} SAH {c / A[0]:.2f} | bounce {traverse(wf, sec)} | primary {traverse(wf, pr[:, :600])}  ({time.time() - t0:.0f}s)', flush=True)
