import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cmc_fluid_solver_amd import capi, grids
n=int(sys.argv[1]) if len(sys.argv)>1 else 256
d=int(sys.argv[2]) if len(sys.argv)>2 else 2
g = grids.box_with_obstacle(n, h=1.0/(n-1))
params = capi.fluid_params(np.float32, 200.0, 0.72, 1.4)
base = [np.ascontiguousarray(a, np.float32) for a in (g.vx, g.vy, g.vz, g.T)]
cur = grids.perturb(base, seed=11); tmp = grids.perturb(base, seed=12)
outs=[]
for kernel in (capi.SWEEP_LINE, capi.SWEEP_PIPE, capi.SWEEP_PIPE, capi.SWEEP_PIPE):
    s = capi.Solver(g, params, np.float32); s.set_option(capi.OPT_SWEEP_KERNEL, kernel)
    s.upload_layer(capi.LAYER_CUR, cur); s.upload_layer(capi.LAYER_TEMP, tmp)
    s.sweep(d, 0.1, capi.LAYER_CUR, capi.LAYER_TEMP, capi.LAYER_NEXT, merge=True)
    outs.append(s.download_layer(capi.LAYER_NEXT) + s.download_layer(capi.LAYER_TEMP)); s.close()
ref=outs[0]
for r,o in enumerate(outs[1:]):
    for k,(a,b) in enumerate(zip(o,ref)):
        idx=np.argwhere(a!=b)
        if len(idx):
            print("run",r,"array",k,"n",len(idx),"i",sorted(set(idx[:,0]))[:12],"j",sorted(set(idx[:,1]))[:24],"k",sorted(set(idx[:,2]))[:40])
            print("    sample",[(tuple(int(v) for v in ix), float(a[tuple(ix)]), float(b[tuple(ix)])) for ix in idx[:3]])
print("done")
