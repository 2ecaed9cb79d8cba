import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TQDM_DISABLE", "1")
import numpy as np, torch
from montecarlo_amd import synth
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
m = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10)
ml2p_, lens_, syms_ = m._dev.enumerate(28.0, 4, 256)
test = m._flat.decode(syms_[:1000000], lens_[:1000000])
scored = m.parse_file(synth.as_file(test))
apps = scored.count
def T(tag, f):
    t=time.perf_counter(); r=f(); print(f"{tag:30s} {(time.perf_counter()-t)*1e3:8.2f} ms", flush=True); return r
for rep in range(3):
    c = T("cumsum", lambda: np.cumsum(apps, dtype=np.int64))
    total = int(c[-1])
    d = T("c / float(total)", lambda: c / float(total))
    e = T("d * 100", lambda: d * 100)
    f = T("c.astype(f8)", lambda: c.astype(np.float64))
    g = T("true_divide f8", lambda: f / float(total))
    print(total, c.dtype, c[:3], d[:3])
from montecarlo_amd.lib4mc.MonteCarloLib import MonteCarloLib
import gc
gc.callbacks.append(lambda ph, info: print('GC', ph, info, time.perf_counter(), flush=True) if info.get('generation')==2 else None)
keys = m.sample(10_000_000)
mc = MonteCarloLib(keys)
for rep in range(2):
    c = T("B cumsum", lambda: np.cumsum(apps, dtype=np.int64))
    d = T("B c / float(total)", lambda: c / float(total))
    e = T("B d * 100", lambda: d * 100)
x = torch.from_numpy(scored.ml2p).cuda()
pos, cnt = mc.device_rank.query(x)
torch.cuda.synchronize()
for rep in range(2):
    c = T("C cumsum", lambda: np.cumsum(apps, dtype=np.int64))
    d = T("C c / float(total)", lambda: c / float(total))
    e = T("C d * 100", lambda: d * 100)
from montecarlo_amd.engine import device_guess_numbers
r_, e_, _ = device_guess_numbers(pos, cnt, True)
torch.cuda.synchronize()
for rep in range(2):
    c = T("D cumsum", lambda: np.cumsum(apps, dtype=np.int64))
    d = T("D c / float(total)", lambda: c / float(total))
    e = T("D d * 100", lambda: d * 100)
h = r_.cpu().numpy()
for rep in range(2):
    c = T("E cumsum", lambda: np.cumsum(apps, dtype=np.int64))
    d = T("E c / float(total)", lambda: c / float(total))
    e = T("E d * 100", lambda: d * 100)
for rep in range(3):
    t=time.perf_counter(); gc_ = mc.ml2p_iter2gc(scored); print(f"ml2p_iter2gc {(time.perf_counter()-t)*1e3:.2f} ms", flush=True)
