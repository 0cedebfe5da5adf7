"""Dev: LDS bank-conflict model of the staged u8 gather's tap reads (ds_read2_b32 = two b32 reads, 2 x 32 lanes, bank = dword address
mod 32; MI355X_MICROARCH.md "LDS") on the bench rotation (12 degrees, scale 0.9, 3840 x 2160): cycles relative to conflict-free for
the box pitch, padded / xor layouts, and every pitch residue mod 32.  Measured (profiles/r04x): 2.2x with the box pitch."""
import numpy as np, math
W,H=3840,2160
ang=math.radians(12.0); s=0.9
# forward M = rotation about center scale s ; inverse maps dst->src
a=s*math.cos(ang); b=s*math.sin(ang)
cx,cy=W/2,H/2
M=np.array([[a,b,(1-a)*cx-b*cy],[-b,a,b*cx+(1-a)*cy],[0,0,1]])
Mi=np.linalg.inv(M)
def tile(bx,by,TW=64,TH=32):
    xs=bx*TW+np.arange(TW); ys=by*TH+np.arange(TH)
    X,Y=np.meshgrid(xs,ys)
    sx=Mi[0,0]*X+Mi[0,1]*Y+Mi[0,2]; sy=Mi[1,0]*X+Mi[1,1]*Y+Mi[1,2]
    return np.floor(sx).astype(int),np.floor(sy).astype(int)
def cycles(addr_groups):
    # addr_groups: array [ngroups, lanes] of dword addresses; cycles per group = max over banks of distinct addrs
    tot=0
    for g in addr_groups:
        banks={}
        for a in set(g.tolist()):
            banks.setdefault(a%32,set()).add(a)
        tot+=max(len(v) for v in banks.values())
    return tot
def sim(layout, lane_map="quad", ntiles=40, seed=0):
    rng=np.random.default_rng(seed); tot=0; ideal=0
    for _ in range(ntiles):
        bx=rng.integers(8,50); by=rng.integers(10,55)
        xi,yi=tile(bx,by)
        xmin,ymin=xi.min(),yi.min(); pitch=((xi.max()+2-xmin+3)//4)*4
        c=xi-xmin; r=yi-ymin
        la=layout(r,c,pitch)
        la1=layout(r,c+1,pitch); lb=layout(r+1,c,pitch); lb1=layout(r+1,c+1,pitch)
        for w in range(8):
            for j in range(4):
                if lane_map=="quad":
                    rows=np.arange(4*w,4*w+4); cols=4*np.arange(16)+j
                    sel=lambda A: A[np.ix_(rows,cols)].reshape(2,32)
                else: # lane = consecutive pixels: wave covers rows 4w..4w+3?? j-th row, 64 px
                    sel=lambda A: A[4*w+j,:].reshape(2,32)
                for A in (la,la1,lb,lb1):
                    tot+=cycles(sel(A)); ideal+=2
    return tot/ideal
lin=lambda r,c,p: r*p+c
def padded(k):
    return lambda r,c,p: r*(p+p//32*k+k)+c+(c//32)*k
print("linear quad", sim(lin))
for k in (1,3): print("pad",k, sim(padded(k)))
print("linear consecutive-lanes", sim(lin,"row"))
# odd pitch
print("pitch+1", sim(lambda r,c,p: r*(p+1)+c))
print("xor swz", sim(lambda r,c,p: r*p+(c^((c>>5)&3))))
print("--- pitch search (quad mapping)")
for k in range(0,32):
    print(k, round(sim(lambda r,c,p,k=k: r*(((p+31)//32)*32+k)+c, ntiles=12),3))
