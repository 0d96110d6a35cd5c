import os, sys, torch
ROOT="/root/repo"
for sub in ("oracle","ipp-marl_amd"): sys.path.insert(0, os.path.join(ROOT, sub))
sys.path.insert(0, ROOT)
from bench import bench_params
from ippmarl.vec_env import VecEnv, POLICY_UNIFORM
class A: envs, agents, grid, actions, terrain, episode_comm_range = 1024, 4, 256, None, "random_field", False
env = VecEnv(bench_params(A), 1024, philox_seed=3, terrain="random_field", track_area=False)
env.reset(list(range(1,1025)))
for t in range(env.d.budget+1): env.steps(t, policy=POLICY_UNIFORM, features=False)
torch.cuda.synchronize()
loc=(env.rows_view(env.local)!=0); glo=(env.rows_view(env.glob)!=0)     # (rows_view: the [.., gx, gy] picture whatever the storage layout)
print("written fraction local", float(loc.float().mean()), "global", float(glo.float().mean()))
ws=env.ws.cpu().numpy()
import numpy as np
def box(ws, a, b):
    bx, by = ws[..., a].astype(np.int64)&0xFFFFFFFF, ws[..., b].astype(np.int64)&0xFFFFFFFF
    x0,x1,y0,y1 = bx&0xFFFF, bx>>16, by&0xFFFF, by>>16
    return x0,x1,y0,y1
fx0,fx1,fy0,fy1 = box(ws,6,7); sx0,sx1,sy0,sy1 = box(ws,14,15)
ux0=np.where(sx1>sx0, np.minimum(np.where(fx1>fx0,fx0,1<<20), sx0), fx0); ux1=np.maximum(fx1,sx1)
uy0=np.where(sy1>sy0, np.minimum(np.where(fy1>fy0,fy0,1<<20), sy0), fy0); uy1=np.maximum(fy1,sy1)
area=(np.maximum(ux1-ux0,0)*np.maximum(uy1-uy0,0))/65536.0
print("box fraction local", area[:,:4].mean(), "global", area[:,4].mean())
# finer: 32-row x 64-col blocks dirty
l=loc.view(1024,4,8,32,4,64).any(dim=3).any(dim=-1); g=glo.view(1024,8,32,4,64).any(dim=2).any(dim=-1)
print("32x64 blocks dirty: local", float(l.float().mean()), "global", float(g.float().mean()))
l=loc.view(1024,4,16,16,8,32).any(dim=3).any(dim=-1); g=glo.view(1024,16,16,8,32).any(dim=2).any(dim=-1)
print("16x32 blocks dirty: local", float(l.float().mean()), "global", float(g.float().mean()))
l=loc.view(1024,4,256,4,64).any(dim=-1); g=glo.view(1024,256,4,64).any(dim=-1)
print("row x 64-col segments dirty: local", float(l.float().mean()), "global", float(g.float().mean()))
# round 6: what a fill by (map, slab of R rows, ONE column interval) would write -- the hull of the written columns per slab
def slab_interval_fraction(w, R):
    # w: bool [..., 256, 256]
    s = w.reshape(*w.shape[:-2], 256 // R, R, 256).any(dim=-2)            # [..., slabs, cols]
    cols = torch.arange(256, device=w.device)
    lo = torch.where(s, cols, torch.full_like(cols, 1 << 20)).amin(dim=-1)
    hi = torch.where(s, cols + 1, torch.zeros_like(cols)).amax(dim=-1)
    return float((hi - lo).clamp_min(0).float().mean() / 256.0)
for R in (64, 32, 16, 8):
    print(f"{R}-row slab x one column interval: local", slab_interval_fraction(loc, R), "global", slab_interval_fraction(glo, R))
