import sys, time
sys.path.insert(0, "ipp-marl_amd"); sys.path.insert(0, ".")
import torch
from ippmarl.vec_env import SplitVecEnv
s = [torch.cuda.Stream() for _ in range(12)]
cyc = 1 << 20
for _ in range(4):
    r, serial = SplitVecEnv._side_by_side(s[0], s[0], cyc)
    print("cycles", cyc, "serial pair", round(serial * 1e3, 3), "ms ratio same-stream", round(r, 2))
    if serial >= 4e-4: break
    cyc <<= 2
print("pairs (0, k):", [round(SplitVecEnv._side_by_side(s[0], s[k], cyc)[0], 2) for k in range(1, 12)])
print("pairs (1, k):", [round(SplitVecEnv._side_by_side(s[1], s[k], cyc)[0], 2) for k in range(2, 12)])
