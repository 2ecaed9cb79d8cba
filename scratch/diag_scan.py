import sys, numpy as np, torch
sys.path.insert(0, '.')
from montecarlo_amd.engine import DeviceRank
torch.manual_seed(0)
n = 10_000_000
keys = (torch.randn(n, dtype=torch.float64, device='cuda') * 8 + 40).abs().contiguous()
rk = DeviceRank(keys)
srt = rk.keys_sorted
w = torch.exp2(srt - np.log2(n))
tc = torch.cumsum(w, 0)
s = srt.cpu().numpy()
wl = np.exp2(s - np.log2(n))
print("exp2 torch vs numpy max rel", float(np.max(np.abs(w.cpu().numpy() - wl) / wl)))
ref = np.cumsum(wl.astype(np.longdouble))
mine = rk.positions.cpu().numpy()
print("mine  vs longdouble:", float(np.max(np.abs(mine - ref) / ref)))
print("torch vs longdouble:", float(np.max(np.abs(tc.cpu().numpy() - ref) / ref)))
print("np64  vs longdouble:", float(np.max(np.abs(np.cumsum(wl) - ref) / ref)))
