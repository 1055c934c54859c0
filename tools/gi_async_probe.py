#!/usr/bin/env python3
"""Asynchronous ReSTIR GI frames at 4K on the 1 M-triangle hall: host enqueue time per frame against wall time per frame (are the frames
really asynchronous?) and the per-part kernel times of the last frames.   usage: python tools/gi_async_probe.py   (FYPRT_LIB=<alternative libfyprt.so>)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from fypraytracer_amd import capi, scenes
W,H=3840,2160
sc=scenes.hall_scene(); cam=scenes.hall_camera(W,H)
import os
if os.environ.get('FYPRT_LIB'): capi._lib = capi.load_library(os.environ['FYPRT_LIB'])
ctx=capi.Context(0); ctx.resize(W,H); ctx.upload_scene(sc); ctx.set_camera(cam)
st=capi.Settings(technique=capi.RESTIR_GI, light_bounces=2, sky_color=(0,0,0), use_temporal_reuse=1, use_spatial_reuse=1)
for f in range(5):
    st.rand_seed=f+1; ctx.render_async(st)
ctx.synchronize()
t0=time.perf_counter()
N=20
for f in range(N):
    st.rand_seed=6+f; ctx.render_async(st)
t1=time.perf_counter()
ctx.synchronize()
t2=time.perf_counter()
print("enqueue ms/frame", (t1-t0)/N*1e3, "wall ms/frame", (t2-t0)/N*1e3)
parts=np.array([ctx.frame_timings(b)[0] for b in range(10)])
print("parts median", np.median(parts,axis=0))
