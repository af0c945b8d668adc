#!/usr/bin/env python
"""cProfile of the host side of the TGN pipeline's fast variant (tools/bench_tgn.py): where the Python time per batch goes."""
import cProfile
import os
import pstats
import runpy
import sys

sys.argv = ['bench_tgn.py', '200', 'fast']
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench_tgn.py'), run_name='__main__')
finally:
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(r'tgm_amd|torch\._C|built-in method torch|ctypes|bench_tgn', 45)
