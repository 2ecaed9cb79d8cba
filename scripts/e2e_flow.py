#!/usr/bin/env python3
"""Wall time of the reference's driver flow (backwords_simulator.py:80-94) at config-2 size through
the drop-in classes: train -> sample -> MonteCarloLib -> parse_file -> ml2p_iter2gc -> write2."""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TQDM_DISABLE", "1")
import torch
from montecarlo_amd import synth
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
from montecarlo_amd.lib4mc.MonteCarloLib import MonteCarloLib

N, T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000, int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
t = [time.perf_counter()]
def lap(name):
    torch.cuda.synchronize(); t.append(time.perf_counter()); print(f"{name:28s} {t[-1]-t[-2]:8.3f} s", flush=True)
train = synth.zipf_corpus(100000); test = synth.zipf_corpus(T, oov_frac=0.01, stream=1); lap("synthetic corpora")
m = BackWordsMonteCarlo(synth.as_file(train), splitter="", threshold=10); lap("train + compile table")
m._dev; lap("upload")
if len(sys.argv) > 3 and sys.argv[3] == "unique":
    # a test set of T DISTINCT passwords (the model's own most probable strings): the host glue at its worst
    ml2p_, lens_, syms_ = m._dev.enumerate(28.0, 4, 256)          # ~2e6 distinct strings of this model
    test = m._flat.decode(syms_[:T], lens_[:T])
    lap(f"unique test set ({len(test)} distinct strings)")
keys = m.sample(N); lap(f"sample({N})")
mc = MonteCarloLib(keys); lap("MonteCarloLib(keys)")
fd = synth.as_file(test); lap("test file object")
scored = m.parse_file(fd); lap(f"parse_file({T} lines)")
gc = mc.ml2p_iter2gc(scored); lap("ml2p_iter2gc")
buf = io.StringIO(); mc.write2(buf); lap("write2")
print("unique test pwds", len(scored), "output bytes", len(buf.getvalue()))
t0 = t[-4]
print(f"parse_file + ml2p_iter2gc + write2: {t[-1] - t0:.3f} s; tuples built: "
      f"{scored._rows is not None or gc._rows is not None}", flush=True)
