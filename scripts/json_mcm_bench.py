"""`.mcm`-shaped JSON writing: 60 leaves of 11 x 1000 values (paper scale), leaves encoded side by side vs one after the other
(cache_io.write_json).  Development aid; no GPU needed."""
import importlib,sys,time,json,os,tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
cio=importlib.import_module("code-robchar_amd.cache_io")
rng=np.random.default_rng(0)
obj={a:{f"m{j}":rng.random((11,1000)) for j in range(15)} for a in ("ppo","snob","nmplus","lbfgs")}
p=os.path.join(tempfile.mkdtemp(),"x.mcm")
for rep in range(4):
    t=time.perf_counter(); cio.write_json(obj,p); print("parallel %.2f ms"%(1e3*(time.perf_counter()-t)))
cio._MID_MAX=0
for rep in range(3):
    t=time.perf_counter(); cio.write_json(obj,p); print("serial %.2f ms"%(1e3*(time.perf_counter()-t)))
print(os.cpu_count(), len(os.sched_getaffinity(0)))
