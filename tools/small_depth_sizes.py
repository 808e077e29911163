import os, sys, time
sys.path.insert(0,'/root/repo')
import numpy as np
from spherical_bundle_adjuster_amd import api, synthetic
for n in (512, 1024, 2048, 2560, 4096, 6144, 8192, 12288, 16384, 32768):
    c = synthetic.full_rt(n, seed=123, sigma=2e-4, outlier_fraction=0.02)
    start = np.full((n,2), 6.0)
    row = {"n": n}
    for name, maxn, one in (("one_launch","100000",None), ("resident","100000","0"), ("launch","0","0")):
        os.environ["SBA_RESIDENT_MAX_N"]=maxn
        if one is None: os.environ.pop("SBA_SMALL_ONE_LAUNCH",None)
        else: os.environ["SBA_SMALL_ONE_LAUNCH"]=one
        with api.Problem(0) as p:
            p.upload(c.x1,c.x2,start); p.solve_depths(c.rot_init,c.tran_init)
            ts=[]
            for _ in range(7):
                p.set_depths(start); t0=time.perf_counter(); d,s=p.solve_depths(c.rot_init,c.tran_init); ts.append(time.perf_counter()-t0)
            row[name]=round(float(np.median(ts))*1e6,1); row["passes"]=s.num_evaluations
    print(row, flush=True)
