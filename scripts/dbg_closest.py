import numpy as np, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import raytracer_amd as rt
from raytracer_amd.flatten import flatten_scene
from oracle import binding as ob
import helpers
sc=helpers.small_scenes()["Testing"](); flat=flatten_scene(sc); dev=rt.DeviceScene(flat); orc=ob.OracleScene(flat)
rs=np.random.default_rng(11)
o1,d1=helpers.camera_rays(sc,64,64,50000,rs); o2,d2=helpers.secondary_rays(flat,50000,rs)
o=np.concatenate([o1,o2]); d=np.concatenate([d1,d2])
for trial in range(2):
    g=dev.TraceClosest(o,d,0.001,np.inf,seed=99); r=orc.trace_closest(o,d,0.001,np.inf,seed=99)
    bad=np.nonzero(g["object"]!=r["object"])[0]
    print("trial",trial,"mismatches",len(bad),"of",len(o), "primary bad", (bad<50000).sum())
    for i in bad[:12]:
        print(i, "gpu obj",g["object"][i],"k",g["k"][i],"| ref obj",r["object"][i],"k",r["k"][i])
    print("gpu miss where ref hit:", ((g["object"]==0xFFFFFFFF)&(r["object"]!=0xFFFFFFFF)).sum(), " gpu hit where ref miss:", ((g["object"]!=0xFFFFFFFF)&(r["object"]==0xFFFFFFFF)).sum())
    print("bad idx mod 64 hist:", np.bincount(bad%64,minlength=64))
