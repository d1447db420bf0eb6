import numpy as np, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import raytracer_amd as rt
from raytracer_amd.core import RenderOptions
from raytracer_amd.flatten import flatten_scene
from oracle import binding as ob
import helpers
sc=helpers.small_scenes()["Testing"](); flat=flatten_scene(sc); dev=rt.DeviceScene(flat); orc=ob.OracleScene(flat)

try: print(dev.Info())
except Exception as e: print('no info')
for depth in (0,1,2,10):
  for ls in (1,2):
    opt=RenderOptions(RenderWidth=96,RenderHeight=54,Passes=8,MaxBounceDepth=depth,LightSampleCountHint=ls,Seed=1234)
    job=rt.RenderJob(sc,opt,device_scene=dev); job.Wait(); got,n=job.ReadAccum()
    ref,rst=orc.render(opt,threads=8)
    err=np.abs(got-ref); scale=np.maximum(np.abs(ref),1e-3)
    close=(err<=1e-3*scale+1e-5).all(axis=2)
    st=job.RenderStats
    print(depth,ls,'close %.4f'%close.mean(),'rays',st.RayCount,rst['ray_count'],'shadow',st.ShadowRays,rst['shadow_rays'],'nodes',st.NodeVisits,'prims',st.PrimTests, 'sums', got.sum(), ref.sum())
    job.Dispose()
print("--- no lights")
from raytracer_amd.core import Scene
sc2=Scene(sc.Name, sc.Camera, sc.SceneObjects, [], sc.SkyBox)
flat2=flatten_scene(sc2); dev2=rt.DeviceScene(flat2); orc2=ob.OracleScene(flat2)
for depth in (0,1,10):
    opt=RenderOptions(RenderWidth=96,RenderHeight=54,Passes=8,MaxBounceDepth=depth,LightSampleCountHint=1,Seed=1234)
    job=rt.RenderJob(sc2,opt,device_scene=dev2); job.Wait(); got,n=job.ReadAccum()
    ref,rst=orc2.render(opt,threads=8)
    err=np.abs(got-ref); scale=np.maximum(np.abs(ref),1e-3)
    close=(err<=1e-3*scale+1e-5).all(axis=2)
    st=job.RenderStats
    print(depth,'close %.4f'%close.mean(),'rays',st.RayCount,rst['ray_count'],'nodes',st.NodeVisits,'prims',st.PrimTests, 'sums', got.sum(), ref.sum())
    job.Dispose()
