import numpy as np
def tile_counts(tw, th, A, P=728, N=512, x0=192, y0=172):
    # a generic interior tile at (x0,y0) in the slice; counts per (angle, slot)
    py=(P-N)//2; X0=py+x0; Y0=py+y0
    diag=np.sqrt((tw+1)**2+(th+1)**2); radius=0.5*diag+3; span=int(np.ceil(2*radius))+2; nb=(span+63)//64*64
    cx=X0+0.5*(tw-1); cy=Y0+0.5*(th-1)
    res=[]
    for t in np.pi*np.arange(A)/A:
        c,s=np.cos(-t),np.sin(-t)
        xo=((P-1)-(c*(P-1)-s*(P-1)))/2; yo=((P-1)-(s*(P-1)+c*(P-1)))/2
        jc=c*(cx-xo)+s*(cy-yo); fb=int(np.floor(jc-radius))
        j=(fb+np.arange(nb))[:,None]; i=np.arange(P)[None,:]
        x=c*j - s*i + xo; y=s*j + c*i + yo
        own=(np.floor(x)>=X0)&(np.floor(x)<X0+tw)&(np.floor(y)>=Y0)&(np.floor(y)<Y0+th)&(j>=0)&(j<P)
        cnt=own.sum(1); cnt[span:]=0
        res.append(cnt)
    return np.array(res), nb
for tw,th in ((64,86),(64,103),(32,86),(64,64),(128,43)):
    cnt,nb=tile_counts(tw,th,90)
    live=cnt.sum()
    # (a) mirrored 32-runs
    wa=0
    for a in range(cnt.shape[0]):
        for blk in range(nb//64):
            sl=np.r_[blk*32:blk*32+32, nb-32*(blk+1):nb-32*(blk+1)+32]
            k=cnt[a,sl].max(); wa+= (k+4 if k>0 else 0)*64
    # (b) per angle: 16-slot bands sorted by max, 4 bands per task
    wb=0
    for a in range(cnt.shape[0]):
        bands=cnt[a].reshape(-1,16).max(1); bands=np.sort(bands)[::-1]
        for q in range(0,len(bands),4):
            k=bands[q]; wb+=(k+4 if k>0 else 0)*64
    # (c) all 16-slot bands of all angles sorted (one class), 4 per task
    bands=np.sort(cnt.reshape(cnt.shape[0],-1,16).max(2).ravel())[::-1]
    wc=sum((bands[q]+4 if bands[q]>0 else 0)*64 for q in range(0,len(bands),4))
    # (d) ideal: all rays sorted
    r=np.sort(cnt.ravel())[::-1]
    wd=sum((r[q]+4 if r[q]>0 else 0)*64 for q in range(0,len(r),64))
    print(f"tile {tw}x{th}: nb {nb} live {live} | mirrored runs eff {live/wa:.2f} | sorted bands/angle {live/wb:.2f} | sorted bands all angles {live/wc:.2f} | sorted rays {live/wd:.2f}")
