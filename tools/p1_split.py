import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from fypraytracer_amd import capi, scenes
W,H=1920,1080
sc=scenes.hall_scene(); cam=scenes.hall_camera(W,H)
ctx=capi.Context(0); ctx.resize(W,H); ctx.upload_scene(sc); ctx.set_camera(cam)
ctx.set_tuning(4,128)
def run(**kw):
    st=capi.Settings(technique=7, sky_color=(0,0,0), **kw)
    ctx.reset_frame_index()
    parts=[]
    for f in range(14):
        st.rand_seed=f+1; s=ctx.render(st)
        if f>=4: parts.append(list(s.kernel_ms_part)[:3])
    return np.median(np.array(parts),axis=0).round(4).tolist()
print("full           ", run(use_temporal_reuse=1,use_spatial_reuse=1,light_candidate_count=4))
print("no temporal    ", run(use_temporal_reuse=0,use_spatial_reuse=1,light_candidate_count=4))
print("1 candidate    ", run(use_temporal_reuse=0,use_spatial_reuse=1,light_candidate_count=1))
print("no spatial     ", run(use_temporal_reuse=1,use_spatial_reuse=0,light_candidate_count=4))
print("16 candidates  ", run(use_temporal_reuse=1,use_spatial_reuse=1,light_candidate_count=16))
def run_primary():
    st=capi.Settings(technique=0, sky_color=(0,0,0), light_bounces=0)
    ctx.reset_frame_index(); ks=[]
    for f in range(14):
        st.rand_seed=f+1; s=ctx.render(st)
        if f>=4: ks.append(s.kernel_ms)
    return round(float(np.median(ks)),4)
print("primary ray only (brute force, 0 bounces)", run_primary())
