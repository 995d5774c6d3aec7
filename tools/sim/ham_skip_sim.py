import sys, numpy as np
sys.path.insert(0,'/root/repo')
import bench
from oracle import cref
from photogrammetry_amd import synth
W,H=1920,1080
base=bench.base_frame(W,H,4321)
pairs=cref.gaussian_pairs(0,50,256)
dmap=cref.build_distortion_matrix(W,H,[3e-4,1e-7,0,0,0])
def det(i):
    f=np.roll(base,(i%H,(3*i)%W),axis=(0,1))
    g=cref.gray(cref.apply_distortion(f,dmap)); raw=cref.detect(g,np.float32(0.1)); kept=raw[cref.nms(raw,16)][:4096]
    return cref.brief(g,np.stack([kept["x"],kept["y"]],1),pairs)
def bits(d): return np.unpackbits(d.view(np.uint8),axis=1).astype(np.int16)*2-1
for (a,b) in ((0,1),(0,20),(0,63)):
    A,B=bits(det(a)),bits(det(b))
    dot=(A@B.T)//2   # [rows][cols] dot/2 in [-128,128]
    n1,n2=dot.shape
    nt=n2//32
    # key order: (dot, smaller ct) -> value = dot*256 + (127-ct)
    tot=skip=0; skip_elem=0
    # a wave tile: 32 rows x 32 cols; lane (r,h): column r, rows g -> (g&3)+8*(g>>2)+4h
    rowsets=[[(g&3)+8*(g>>2)+4*h for g in range(16)] for h in range(2)]
    for rt in range(0,(n1//32)):
        blk=dot[rt*32:(rt+1)*32]            # 32 rows
        best=np.full(32,-10**9)             # per row best so far (exact per row)
        # per-lane bests: lane holds rbest per (row, its column class r): class-specific best!
        lb=np.full((32,32),-10**9)          # [row][r] best over tiles for column class r
        for ct in range(nt):
            v=blk[:,ct*32:(ct+1)*32]*256+(127-ct)    # [row][r]
            # lane (r,h): xm = max over its 16 rows of v[row][r]; floor = min over its 16 rows of lb[row][r]
            upd_any=False
            for h in range(2):
                rs=rowsets[h]
                xm=v[rs,:].max(axis=0)          # per r
                fl=lb[rs,:].min(axis=0)
                if (xm>fl).any(): upd_any=True
            tot+=1
            if not upd_any: skip+=1
            lb=np.maximum(lb,v)
    print((a,b),"n",n1,n2,"wave-uniform skip rate %.3f"%(skip/tot))
