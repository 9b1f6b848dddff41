#!/usr/bin/env python3
"""What would the walk cost if the triangle tests were somebody else's work?  (round 3: the premise check that stopped a "decoupled triangle
queue" redesign before it was written.)  The bounce-1 batch of the bench frame, tiled to 46 M rays, through rt3_trace_rays; run once with the
product's librt3.so and once with a throw-away build of rt3_kernels.hip compiled with -DRT3_EXP_SKIP_TRIS (a leaf is fetched and popped, no
triangle is tested: 39 % fewer vector instructions in an iteration that visits a leaf lane).  Result on the box: 6.02 ps per lane-iteration
with the tests, 5.17 without -- the iteration is a dependent fetch first and ~235 vector instructions second (elasticity 0.36), so taking the
tests out of the loop and running them in iterations of their own (3.5 more wave-iterations per 64 rays against 19.8 / 0.9 = 22 now) cannot pay.
Test infrastructure (uses the oracle): lives under tests/.   RT3_LIBRARY=<variant .so> python tests/experiments/exp_skip_tris.py"""
import sys, math, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import orc
from raytracer3_amd import scenes, assets
from raytracer3_amd.render_graph import Context
mesh=scenes.atrium(1.0)
osc=orc.Scene(mesh)
W,H=1920,1080
g=orc.camera_gconst(width=W,height=H,**scenes.ATRIUM_CAMERA); g.samples=1; g.bounces=4; g.pad[0]=12; g.blendfactor=1.0
gb,depth=osc.gbuffer(g,threads=16)
b1=osc.bounce1_rays(g,gb,depth)
b1=np.ascontiguousarray(np.tile(b1,(1,24)))  # 46 M rays: a launch long enough for the dynamic pool to matter
ctx=Context(0); ctx.upload_mesh(mesh); ctx.build_accel()
r=ctx.trace_rays(b1,counts=True)
t,u,v,p,cn,ct,ms=r
r2=ctx.trace_rays(b1,repeat=5)
print('rays',b1.shape[1],'hit frac',(p!=0xFFFFFFFF).mean(),'nodes/ray',cn.mean(),'tris/ray',ct.mean(),'kernel ms',r2[4],'ps per step',r2[4]*1e-3/(b1.shape[1]*(cn.mean()+ct.mean()))*1e12, 'ps per node step', r2[4]*1e-3/(b1.shape[1]*cn.mean())*1e12)
