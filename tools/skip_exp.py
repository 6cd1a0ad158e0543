"""Timing experiment (results are garbage on purpose): run bench.py with one C-ABI entry point turned into a no-op, to see
how much of that kernel family's time is NOT hidden behind other work.  python tools/skip_exp.py sempyr_affine_act [bench args]"""
import os, runpy, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
skip = set(sys.argv[1].split(',')) if sys.argv[1] != 'none' else set()
from semantic_pyramid_for_image_generation_amd import ops, optim


class Proxy:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        if name in skip:
            return lambda *a: 0
        fn = getattr(self._lib, name)
        kinds = [int(k.split(':')[1]) for k in skip if k.startswith(name + ':')]      # name:K skips calls whose first argument is K
        if kinds:
            return lambda *a: 0 if a[0] in kinds else fn(*a)
        return fn


ops.L = Proxy(ops.L)
optim.L = Proxy(optim.L)
sys.argv = ['bench.py'] + sys.argv[2:]
runpy.run_path(os.path.join(R, 'bench.py'), run_name='__main__')
