import sys, os, random
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["TQDM_DISABLE"]="1"
import numpy as np
from montecarlo_amd import synth
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
from walker import CompiledWalker
m = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10)
w = CompiledWalker(m._flat)
ats = []
orig = w.lookup
def lookup(goff, gbits, r):
    K = 1 << gbits
    ats.append(goff + min(int(r*K), K-1))
    return orig(goff, gbits, r)
w.lookup = lookup
rnd = random.Random(1)
for _ in range(4000):
    w.sample(rnd.random)
a = np.array(ats)
print("draws", len(a), "table recs", len(w.draw))
for kb in (8, 12, 19, 30, 38, 60, 76, 100, 150, 300, 1000):
    print(kb, "KB:", round(float((a < kb*1024//16).mean()), 3))
