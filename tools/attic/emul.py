import sys, os, math, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_ops import _affines
M=_affines()['big_rigid'].numpy().astype(np.float32)
sd=(12,10,9); gd=(11,12,10)
TX,TY,TZ=8,8,30
def g_of(ui,uj,uk):
    return np.array([M[r,0]*ui+M[r,1]*uj+M[r,2]*uk+M[r,3] for r in range(3)],dtype=np.float64)
conf=0
for x0 in range(0,sd[0],TX):
  for y0 in range(0,sd[1],TY):
    z0=0
    ex,ey,ez=min(TX,sd[0]-x0),min(TY,sd[1]-y0),min(TZ,sd[2]-z0)
    lo=np.array([x0-1,y0-1,z0-1],float); hi=np.array([x0+ex,y0+ey,z0+ez],float)
    segs=[]
    for ui in range(gd[0]):
      for uj in range(gd[1]):
        ks=[uk for uk in range(gd[2]) if np.all(g_of(ui,uj,uk)>=lo) and np.all(g_of(ui,uj,uk)<hi)]
        if ks: segs.append((ui,uj,ks[0],len(ks), ks))
    # each segment alone: check planes
    for (ui,uj,k0,n,ks) in segs:
        assert ks==list(range(k0,k0+n)), ('non contiguous',ui,uj,ks)
        lz=[int(math.floor(g_of(ui,uj,k)[2])) for k in ks]
        cells=[tuple(int(math.floor(v)) for v in g_of(ui,uj,k)) for k in ks]
        dup=[i>0 and lz[i]==lz[i-1] for i in range(n)]
        for turn in (0,1):
            lanes=[i for i in range(n) if dup[i]==bool(turn)]
            seen={}
            for i in lanes:
                cx,cy,cz=cells[i]
                for dx in (0,1):
                  for dy in (0,1):
                    key=(cx+dx,cy+dy,cz)
                    if key in seen: conf+=1; print('conflict grp1 tile',x0,y0,'row',ui,uj,'lanes',seen[key],i,'lz',lz, 'dup',dup); 
                    seen[key]=i
print('conflicts',conf)
# --- row_sep as splat_safety computes it, then brute-force verify
c=M[:3,2].astype(float)
def disjoint(di,dj):
    v=di*M[:3,0].astype(float)+dj*M[:3,1].astype(float)
    tau=np.arange(-48,48,0.02)
    m=np.max(np.abs(v[:,None]+tau[None,:]*c[:,None]),axis=0)
    return m.min()>=2.03
row_sep=None
for n in range(1,7):
    ok=all(disjoint(di,dj) for di in range(-n-3,n+4) for dj in range(-n-3,n+4) if max(abs(di),abs(dj))>=n)
    if ok: row_sep=n; break
print('row_sep',row_sep,'c',c)
# brute force: rows (0,0) vs (di,dj) for |.|>=row_sep up to 12: any lanes k,k' in 0..40 with overlapping 2x2x1 footprint in same plane?
bad=0
if row_sep:
  for di in range(-12,13):
    for dj in range(-12,13):
      if max(abs(di),abs(dj))<row_sep: continue
      for k in range(0,40):
        ga=np.floor(g_of(20,20,k))
        for k2 in range(0,40):
          gb=np.floor(g_of(20+di,20+dj,k2))
          d=np.abs(ga-gb)
          if d[2]==0 and d[0]<=1 and d[1]<=1: bad+=1
print('pair conflicts (grp1 same plane overlap)',bad)
# --- emulate pairing + turns over each tile; detect conflicts within (turn, group)
def emu(x0,y0):
    z0=0
    ex,ey,ez=min(TX,sd[0]-x0),min(TY,sd[1]-y0),min(TZ,sd[2]-z0)
    lo=np.array([x0-1,y0-1,z0-1],float); hi=np.array([x0+ex,y0+ey,z0+ez],float)
    segs=[]
    for ui in range(gd[0]):
      for uj in range(gd[1]):
        ks=[uk for uk in range(gd[2]) if np.all(g_of(ui,uj,uk)>=lo) and np.all(g_of(ui,uj,uk)<hi)]
        if ks: segs.append((ui,uj,ks))
    nr=len(segs); npair=(nr+1)//2; nconf=0
    for p in range(npair):
        A=segs[p]; B=segs[p+npair] if p+npair<nr else None
        solo = B is not None and max(abs(A[0]-B[0]),abs(A[1]-B[1]))<row_sep
        lanes=[]
        for half,S in ((0,A),(1,B)):
            if S is None: continue
            prev=None
            for i,k in enumerate(S[2]):
                cell=tuple(int(math.floor(v)) for v in g_of(S[0],S[1],k))
                dup = prev is not None and cell[2]==prev
                prev=cell[2]
                turn=(2 if (solo and half) else 0)+(1 if dup else 0)
                lanes.append((turn,cell,half,i))
        for turn in range(4):
            for grp in (0,1):
                seen={}
                for (t,cell,half,i) in lanes:
                    if t!=turn: continue
                    for dx in (0,1):
                      for dy in (0,1):
                        key=(cell[0]+dx,cell[1]+dy,cell[2]+grp)
                        if key in seen: nconf+=1; print('CONFLICT tile',x0,y0,'pair',p,'turn',turn,'grp',grp,seen[key],(half,i),'rows',A[:2],B[:2] if B else None,'solo',solo)
                        seen[key]=(half,i)
    return nconf,nr
tot=0
for x0 in range(0,sd[0],TX):
  for y0 in range(0,sd[1],TY):
    n,nr=emu(x0,y0); tot+=n; print('tile',x0,y0,'nr',nr,'conf',n)
print('total',tot)
