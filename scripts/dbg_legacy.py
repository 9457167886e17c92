import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
be = importlib.import_module("code-robchar_amd.backend")
np.random.seed(7)
got = be.legacy_normal_periods(1, 100001, 0, [1.0]).cpu().numpy()[0]
np.random.seed(7)
want = np.random.normal(size=100001)
rel = np.abs(got - want) / np.abs(want)
i = np.argsort(rel)[-10:]
for j in i: print(j, got[j], want[j], rel[j] / 2.2e-16, "ulp")
print("hist ulp:", np.histogram(rel / 2.2e-16, bins=[0, 0.5, 1.5, 2.5, 4.5, 8.5, 16.5, 1e9])[0])
# reconstruct r2 for the worst: pair index
