"""Developer study: LDS bank conflicts of the planned forward's gathers (ds_read_b64, slice pairs) under different
per-lane row alignments and row pitches, counted with the gfx950 rule (two 32-lane groups per instruction, 8-byte slot =
index mod 32, identical addresses broadcast, cycles = largest number of distinct addresses on one slot).

    python tools/sim_lds_conflicts.py [angles]      (128 x 128 slice, P = 184)

"current" = what rotate_fwd_first_kernel encodes (every ray starts at its own first live row, pitch == 1 mod 32, x
mirrored for the opposite-sign class); "skew k" = lanes delayed by round(alpha * lane) rows so that consecutive lanes'
taps step by one slot, with pitch == k mod 32.  Measured counterpart: tools/collect_sq.sh (SQ_LDS_BANK_CONFLICT)."""
import sys

import numpy as np

N = 128
P = int(np.ceil((np.sqrt(np.float64(2 * N * N)) + 2) / 2) * 2)
pad = (P - N) // 2
A = int(sys.argv[1]) if len(sys.argv) > 1 else 20
theta = np.pi * np.arange(0, 180, 180 // A)[:A] / 180


def transforms(theta, H, W):
    """tfa.image.rotate's rows for angle -theta (fp32), as ctpvae_rotate_transforms_f32 builds them."""
    ang = (-theta).astype(np.float32)
    c, s = np.cos(ang.astype(np.float64)).astype(np.float32), np.sin(ang.astype(np.float64)).astype(np.float32)
    w1, h1 = np.float32(W - 1), np.float32(H - 1)
    xo = (w1 - (c * w1 - s * h1)) / np.float32(2)
    yo = (h1 - (s * w1 + c * h1)) / np.float32(2)
    z = np.zeros_like(c)
    return np.stack([c, -s, xo, s, c, yo, z, z], axis=1).astype(np.float32)


T = transforms(theta, P, P)
nJB=(P-(P>>1)+31)//32
rows=np.arange(P,dtype=np.float32)
def lane_to_bin(jb,lane):
    c=P>>1
    return np.where(lane<32, c-32*(jb+1)+lane, c+32*jb+(lane-32))
def xy_for(a,j):
    t=T[a]; t0,t1,t2,t3,t4,t5=[np.float32(v) for v in t[:6]]
    x=(t0*np.float32(j)+t1*rows)+t2; y=(t3*np.float32(j)+t4*rows)+t5
    rnd=lambda v: np.where(v>=0,np.floor(v+np.float32(0.5)),np.ceil(v-np.float32(0.5))).astype(np.int64)
    ix=rnd(x)-pad; iy=rnd(y)-pad
    ok=(ix>=0)&(ix<N)&(iy>=0)&(iy<N)
    return ix,iy,ok
cache={}
def block(a,jb):
    if (a,jb) not in cache:
        js=lane_to_bin(jb,np.arange(64))
        X=np.zeros((64,P),np.int64);Y=np.zeros((64,P),np.int64);OK=np.zeros((64,P),bool)
        for l in range(64):
            if 0<=js[l]<P: X[l],Y[l],OK[l]=xy_for(a,js[l])
        cache[(a,jb)]=(X,Y,OK)
    return cache[(a,jb)]
def cost(policy):
    cyc=0;grp=0;instr=0
    for a in range(A):
        for jb in range(nJB):
            X,Y,OK=block(a,jb)
            has=OK.any(1)
            if not has.any(): continue
            first=np.where(has,OK.argmax(1),P); last=np.where(has,P-1-OK[:,::-1].argmax(1),-1)
            start,k,mirror=policy(a,jb,first,last,has)
            pitch={0:128,1:129,-1:159}[k]
            nsteps=int(np.max(np.where(has,last-start+1,0))); ng=(nsteps+7)//8; grp+=ng
            for n in range(ng*8):
                r=start+n; rc=np.clip(r,0,P-1); ar=np.arange(64)
                ok=(r>=0)&(r<P)&OK[ar,rc]
                xx=X[ar,rc]; xx=np.where(mirror,N-1-xx,xx)
                v=np.where(ok,Y[ar,rc]*pitch+xx,N*pitch)
                for h in (slice(0,32),slice(32,64)):
                    vv=np.unique(v[h]); cyc+=np.bincount(vv%32,minlength=32).max()
                instr+=1
    return cyc,grp,instr
def cur(a,jb,first,last,has):
    t=T[a]; plus=(t[0]>=0)==(t[3]>=0)
    return np.where(has,first,0),1,(not plus)
def skew(k):
    def f(a,jb,first,last,has):
        t=T[a].astype(np.float64); ux,uy,vx,vy=t[0],t[3],t[1],t[4]
        den=vx+k*vy
        lanes=(np.arange(64)%32).astype(np.float64)
        best=None
        for target in (1.0,-1.0):
            if abs(den)<1e-3:
                al=0.0
            else:
                al=(target-(ux+k*uy))/den
            d=np.round(al*lanes)
            # start_j = d_j + c, with c = min over live lanes (first_j - d_j), separately per half
            start=np.zeros(64,np.int64)
            for h in (slice(0,32),slice(32,64)):
                if has[h].any():
                    c=np.min((first[h]-d[h])[has[h]]); start[h]=(d[h]+c)
            steps=np.max(np.where(has,last-start+1,0))
            if best is None or steps<best[0]: best=(steps,start)
        return best[1],k,False
    return f
print("A",A)
for name,pol in [("current",cur),("skew k=0",skew(0)),("skew k=1",skew(1)),("skew k=-1",skew(-1))]:
    c=cost(pol); print("%-10s cycles %6d groups %4d instr %5d cyc/instr %.3f"%(name,c[0],c[1],c[2],c[0]/c[2]))
print("per angle: current | k=0 | k=1 | k=-1  (cycles, groups)")
A_all=A
for a0 in range(A_all):
    res=[]
    for pol in (cur,skew(0),skew(1),skew(-1)):
        cyc=0;grp=0
        # restrict to one angle
        def one(policy,a=a0):
            c=0;g=0
            for jb in range(nJB):
                X,Y,OK=block(a,jb); has=OK.any(1)
                if not has.any(): continue
                first=np.where(has,OK.argmax(1),P); last=np.where(has,P-1-OK[:,::-1].argmax(1),-1)
                start,k,mirror=policy(a,jb,first,last,has)
                pitch={0:128,1:129,-1:159}[k]
                nsteps=int(np.max(np.where(has,last-start+1,0))); ng=(nsteps+7)//8; g+=ng
                for n in range(ng*8):
                    r=start+n; rc=np.clip(r,0,P-1); ar=np.arange(64)
                    ok=(r>=0)&(r<P)&OK[ar,rc]
                    xx=X[ar,rc]; xx=np.where(mirror,N-1-xx,xx)
                    v=np.where(ok,Y[ar,rc]*pitch+xx,N*pitch)
                    for h in (slice(0,32),slice(32,64)):
                        vv=np.unique(v[h]); c+=np.bincount(vv%32,minlength=32).max()
            return c,g
        res.append(one(pol))
    print("%5.1f deg "%np.degrees(theta[a0])," | ".join("%5d %3d"%r for r in res))
