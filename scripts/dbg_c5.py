import numpy as np, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import raytracer_amd as rt
from raytracer_amd import builtin_scenes as bs
from raytracer_amd.flatten import flatten_scene
from oracle import binding as ob
import helpers
sc=bs.ManySpheresManyLights(); flat=flatten_scene(sc); dev=rt.DeviceScene(flat); orc=ob.OracleScene(flat)
print(dev.Info())
rs=np.random.default_rng(24)
o1,d1=helpers.camera_rays(sc,64,64,50000,rs); o2,d2=helpers.secondary_rays(flat,50000,rs)
o=np.concatenate([o1,o2]); d=np.concatenate([d1,d2])
g=dev.TraceClosest(o,d,0.001,np.inf,seed=7); r=orc.trace_closest(o,d,0.001,np.inf,seed=7)
bad=np.nonzero(g["object"]!=r["object"])[0]
import ctypes as C
sph=np.frombuffer((C.c_char*(flat.desc.n_spheres*16)).from_address(C.addressof(flat.desc.spheres.contents)),np.float32).reshape(-1,4)
for i in bad:
    print(i, "o",o[i],"d",d[i])
    print(" gpu",g["object"][i],repr(g["k"][i]),"ref",r["object"][i],repr(r["k"][i]))
    for ob_ in (g["object"][i], r["object"][i]):
        if ob_!=0xFFFFFFFF and ob_<len(sph): print("  sphere",ob_,sph[ob_])
objs=np.frombuffer((C.c_char*(flat.n_objects*36)).from_address(C.addressof(flat.desc.objects.contents)),np.uint32).reshape(-1,9)
def isect(oi, o, d):
    t,idx=objs[oi,0],objs[oi,1]
    c=sph[idx,:3].astype(np.float64); rad=float(sph[idx,3])
    oc=o.astype(np.float64)-c; dd=d.astype(np.float64)
    a=dd@dd; hb=oc@dd; cc=oc@oc-rad*rad; disc=hb*hb-a*cc
    return t,idx,sph[idx],disc,((-hb-np.sqrt(max(disc,0)))/a,(-hb+np.sqrt(max(disc,0)))/a)
for i in bad:
    for ob_ in (g["object"][i], r["object"][i]):
        print("  obj",ob_,isect(ob_,o[i],d[i]))
    aabb=objs[:,3:].view(np.float32)
    for ob_ in (g["object"][i], r["object"][i]):
        print("  aabb",ob_,aabb[ob_])
