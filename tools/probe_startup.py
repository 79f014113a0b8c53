import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, sys, os
t0 = time.perf_counter()
import xenomapper_amd.xenomapper as x
from xenomapper_amd import _ffi
t1 = time.perf_counter()
L = _ffi.lib()
t2 = time.perf_counter()
ctx = x.default_context()
t3 = time.perf_counter()
b = x.default_bamdev()
t4 = time.perf_counter()
s = x.default_stripper()
t5 = time.perf_counter()
print("import %.3f  dlopen %.3f  context %.3f  bamdev %.3f  stripper %.3f" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
