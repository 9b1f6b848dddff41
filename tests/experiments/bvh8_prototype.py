#!/usr/bin/env python3
"""Would eight-wide BVH nodes pay?  (the answer recorded in DESIGN.md section 7, round 2.)
Prototype on the CPU oracle's BINARY tree of the bench scene (LBVH leaf pairs under the binned-SAH top), before writing any kernel:
collapse by surface area to 4 and to 8 slots, walk a sample of primary and diffuse rays with (a) exact distance sorting and (b) the
octant slot order of compressed wide BVHs, and count node visits / triangle tests per ray.  Pure Python, a few minutes.
Result on the atrium: the 8-wide collapse fills 4.08 of 8 slots on average; node visits per diffuse ray 13.8 -> 9.5 (sorted) / 9.7
(octant order) -- a third fewer round trips, but each step tests eight boxes: more vector instructions per ray on a kernel that is
at 71 % of its VALU issue ceiling.  Test infrastructure (uses the oracle): lives under tests/."""
import sys, math, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import orc
from raytracer3_amd import scenes
mesh=scenes.atrium(1.0)
osc=orc.Scene(mesh, leaf_size=2, node_width=2, quantized=0, sah_top=2)
nodes=osc.nodes(); tris=osc.tris()
print(nodes.shape, tris.shape)
nf=nodes.view(np.float32)
N=len(nodes)
box=np.zeros((N,2,6),np.float32)  # child k: mn xyz, mx xyz
box[:,0,:3]=nf[:,0:3]; box[:,0,3]=nf[:,3]; box[:,0,4:6]=nf[:,4:6]
box[:,1,0:2]=nf[:,6:8]; box[:,1,2]=nf[:,8]; box[:,1,3:6]=nf[:,9:12]
ref=nodes[:,12:14].copy()
LEAF=0x80000000
def is_leaf(r): return (r & LEAF)!=0
def area(b):
    e=b[3:6]-b[0:3]; return (e[0]*e[1]+e[1]*e[2])+e[2]*e[0]
def collapse(width):
    """returns dict node -> (list of child boxes, list of child refs) for surviving nodes (reachable from root)"""
    out={}; stack=[0]
    while stack:
        i=stack.pop()
        sl=[(box[i,0],int(ref[i,0])),(box[i,1],int(ref[i,1]))]
        while len(sl)<width:
            best=-1; ba=-1.0
            for k,(b,r) in enumerate(sl):
                if is_leaf(r): continue
                a=area(b)
                if a>ba: ba=a; best=k
            if best<0: break
            b,r=sl[best]
            sl[best:best+1]=[(box[r,0],int(ref[r,0])),(box[r,1],int(ref[r,1]))]
        out[i]=(np.array([b for b,_ in sl]),[r for _,r in sl])
        for _,r in sl:
            if not is_leaf(r): stack.append(r)
    return out
import time
t0=time.time(); w4=collapse(4); w8=collapse(8); print('collapse',time.time()-t0, len(w4), len(w8))
print('avg children 4:',np.mean([len(v[1]) for v in w4.values()]),'8:',np.mean([len(v[1]) for v in w8.values()]))
# octant slot assignment for w8: greedy by cost
def assign_slots(bxs, parent_c):
    n=len(bxs); c=(bxs[:,0:3]+bxs[:,3:6])*0.5-parent_c
    cost=np.zeros((n,8))
    for s in range(8):
        d=np.array([1.0 if s&1 else -1.0, 1.0 if s&2 else -1.0, 1.0 if s&4 else -1.0])
        cost[:,s]=c@d
    # greedy: repeatedly take the max cost (child,slot) pair among unassigned
    slot=[-1]*n; used=[False]*8; cc=cost.copy()
    for _ in range(n):
        k,s=np.unravel_index(np.argmax(cc),cc.shape)
        slot[k]=s; used[s]=True; cc[k,:]=-np.inf; cc[:,s]=-np.inf
    return slot
# rays: primary + secondary diffuse
W,H=1920,1080
g=orc.camera_gconst(width=W,height=H,**scenes.ATRIUM_CAMERA)
rng=np.random.default_rng(1)
xs=rng.integers(0,W,3000); ys=rng.integers(0,H,3000)
pr=orc.primary_rays(g,xs,ys)
t,u,v,p,cn,ct=osc.trace_closest(pr,counts=True)
hit=p!=0xFFFFFFFF
P=(pr[0:3]+pr[3:6]*t)[:,hit]
# diffuse directions about approximate normals: use random hemisphere around -dir reflection (rough)
d=rng.normal(size=P.shape); d/=np.linalg.norm(d,axis=0)
flip=(d*pr[3:6][:,hit]).sum(0)>0; d[:,flip]*=-1
sec=np.concatenate([P+d*1e-3,d,np.full((1,P.shape[1]),1e-3),np.full((1,P.shape[1]),1e5)]).astype(np.float32)
def tri_hit(o,dr,first,cnt,tmin,best):
    nt=0
    for k in range(first,first+cnt):
        tf=tris[k].view(np.float32); v0=tf[0:3]; e1=tf[3:6]; e2=tf[6:9]
        nt+=1
        pv=np.cross(dr,e2); det=e1@pv
        if det==0: continue
        inv=1.0/det; tv=o-v0; uu=(tv@pv)*inv
        if uu<0 or uu>1: continue
        qv=np.cross(tv,e1); vv=(dr@qv)*inv
        if vv<0 or uu+vv>1: continue
        tt=(e2@qv)*inv
        if tt>tmin and tt<best: best=tt
    return best,nt
def traverse(wide, rays, mode, slots=None):
    tot_n=0; tot_t=0
    for j in range(rays.shape[1]):
        o=rays[0:3,j].astype(np.float64); dr=rays[3:6,j].astype(np.float64); tmin=rays[6,j]; best=float(rays[7,j])
        inv=1.0/np.where(np.abs(dr)<1e-20,1e-20,dr)
        octm=(1 if dr[0]>=0 else 0)|(2 if dr[1]>=0 else 0)|(4 if dr[2]>=0 else 0)
        stack=[0]
        while stack:
            r=stack.pop()
            if is_leaf(r):
                best,nt=tri_hit(o,dr,r&0x0FFFFFFF,((r>>28)&7)+1,tmin,best); tot_t+=nt; continue
            bxs,refs=wide[r]; tot_n+=1
            t0=(bxs[:,0:3]-o)*inv; t1=(bxs[:,3:6]-o)*inv
            tn=np.maximum(np.minimum(t0,t1).max(1),tmin); tf=np.minimum(np.maximum(t0,t1).min(1),best)
            h=tn<=tf
            idx=np.nonzero(h)[0]
            if len(idx)==0: continue
            if mode=='sort':
                order=idx[np.argsort(tn[idx],kind='stable')]
            else:
                sl=slots[r]; near=(~octm)&7
                order=sorted(idx,key=lambda k:(sl[k]^near))
            for k in reversed(order): stack.append(refs[k])
    return tot_n/rays.shape[1], tot_t/rays.shape[1]
print('oracle 4-wide counts on primary:',cn.mean(),ct.mean())
sub=pr[:,:1500]; sec=sec[:,:1500]
t0=time.time()
print('4-wide sort  primary',traverse(w4,sub,'sort'),' secondary',traverse(w4,sec,'sort'), time.time()-t0)
print('8-wide sort  primary',traverse(w8,sub,'sort'),' secondary',traverse(w8,sec,'sort'))
slots={i:assign_slots(v[0], (np.minimum.reduce(v[0][:,0:3])+np.maximum.reduce(v[0][:,3:6]))*0.5) for i,v in w8.items()}
print('8-wide octant primary',traverse(w8,sub,'oct',slots),' secondary',traverse(w8,sec,'oct',slots))
