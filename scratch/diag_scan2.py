import sys, numpy as np, torch
sys.path.insert(0, '.')
from montecarlo_amd import synth
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
from montecarlo_amd.engine import DeviceRank
m = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10)
n = 10_000_000
keys = m._dev.sample(n, 99)["ml2p"]; m._dev.check()
rk = DeviceRank(keys)
srt = rk.keys_sorted
s = srt.cpu().numpy()
wl = np.exp2(s - np.log2(n))
ref = np.cumsum(wl.astype(np.longdouble))
mine = rk.positions.cpu().numpy()
tc = torch.cumsum(torch.exp2(srt - np.log2(n)), 0).cpu().numpy()
for name, a in (("mine", mine), ("torch", tc)):
    rel = np.abs(a - ref) / ref
    i = int(np.argmax(rel))
    print(name, float(rel[i]), i, a[i-2:i+3], np.asarray(ref[i-2:i+3], dtype=np.float64), s[i-2:i+3])
print(s[:5], s[-5:], np.isnan(s).sum())
