#!/usr/bin/env python3
"""Config 5 on a tick sequence: how well the previous tick's loop trips predict this tick's, and what dealing the solves over the
wavefronts by them would do to the slowest wavefront (model: phase L in rounds of floor(64 / k) lanes per solve + 120 k ticks of
R and F per trip).  Diagnostic tool: the correlation is 0.16-0.35 and the dealt batch's slowest wavefront is no faster."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")]
import numpy as np, torch
import cilqr_amd
from cilqr_amd import scenes
B,N,M=8192,80,16
p=cilqr_amd.default_params(N)
ts=scenes.TickSequence("static",B,p,N=N,M=M,seed=scenes.SEED0+5)
s=cilqr_amd.Solver(p,max_batch=B,max_horizon=N,max_obstacles=M)
ps=torch.zeros(B,dtype=torch.int32,device="cuda"); s.set_pass_count_buffer(ps.data_ptr())
prev=None
for t in range(8):
    i=ts.inputs()
    r=s.solve_batch(N,i["x0"],i["U"],i["poly"],i["xplan_fl"],i["obs_pose"],i["obs_dim"],None)
    trips=ps.cpu().numpy()+(r["status"]!=0)
    S=8
    w=trips.reshape(-1,S).max(axis=1)
    line="tick %d: trips mean %.2f; wave-max mean %.2f; share at 21: %.2f" % (t,trips.mean(),w.mean(),(trips>=20).mean())
    if prev is not None:
        line+="; corr with previous tick %.2f" % np.corrcoef(prev,trips)[0,1]
        # dealing by the previous tick's trips: sort descending, deal round-robin over 1024 waves
        order=np.argsort(-prev,kind="stable"); nw=B//S
        dealt=np.empty(B,dtype=int); dealt[(np.arange(B)%nw)*S+np.arange(B)//nw]=order
        def wave_time(tr):
            tot=[]
            for wv in tr.reshape(-1,S):
                tt=0
                for tau in range(1,wv.max()+1):
                    k=(wv>=tau).sum(); P=64//k
                    tt+=sum(-(-c//P) for c in (32,32,16))*16.7+120
                tot.append(tt)
            return np.max(tot),np.mean(tot)
        print(line, "| slowest/mean wave (k ticks) as given %s | dealt by previous trips %s" % (wave_time(trips), wave_time(trips[dealt])))
    else:
        print(line)
    prev=trips
    ts.advance(r["X"],r["U"])
