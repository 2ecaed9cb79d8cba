#!/usr/bin/env python3
"""cProfile of scripts/e2e_flow.py (same arguments): where the host time of the driver flow goes."""
import cProfile, io, os, pstats, runpy, sys
here = os.path.dirname(os.path.abspath(__file__))
pr = cProfile.Profile()
pr.enable()
runpy.run_path(os.path.join(here, "e2e_flow.py"), run_name="__main__")
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(70)
print(s.getvalue()[:12000])
