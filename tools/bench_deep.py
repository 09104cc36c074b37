import os,sys,time
sys.path.insert(0,'/root/repo'); os.environ.setdefault("RVCX_DEBUG","1")
import numpy as np, polgen_rvc_amd
from polgen_rvc_amd import _lib
ctx=_lib.Context(0)
g=np.random.default_rng(0)
for (C,H,W) in [(512,101,4),(256,202,8),(128,404,16)]:
    x=g.standard_normal((1,C,H,W)).astype(np.float32); w=(g.standard_normal((C,C,3,3))/np.sqrt(9*C)).astype(np.float32)
    ctx.conv2d3x3(x,w,None,act=2)
    ctx.conv_profile_begin()
    for _ in range(10): ctx.conv2d3x3(x,w,None,act=2)
    pr=ctx.conv_profile_end()
    print(C,H,W,[(p['tile'][:28],p['launches'],round(p['ms']/p['launches']*1e3,1)) for p in pr])
