import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, mods_amd as M
from oracle import pyoracle as O
from mods_amd import synthetic
a,b,H=synthetic.make_pair(rows=240, cols=320, nblobs=420, seed=777)
ctx=M.Context(0); im=ctx.upload(a)
k=O.detect_hessaff(a,O.default_params()); regs=O.detect_affine_regions(k)
for mr,ma in ((1.0,1),(5.1962,5)):
    ref=O.detect_orientation(a,regs,mr_size=mr,max_ang=ma); got=ctx.detect_orientation(im,regs.view(M.REGION),mr_size=mr,max_ang=ma)
    print(mr,ma,len(ref),len(got))
    if len(ref)==len(got):
        for f in ('a11','a12','a21','a22','x','y','s'):
            d=np.abs(ref['det_kp'][f]-got['det_kp'][f]); print(f,(d>0).sum(),d.max())
        ang_r=np.arctan2(ref['det_kp']['a12'],ref['det_kp']['a11']); ang_g=np.arctan2(got['det_kp']['a12'],got['det_kp']['a11'])
        print('angle diffs', np.sort(np.abs(ang_r-ang_g))[-5:])
