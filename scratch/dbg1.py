import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, torch, common
from octproz_amd import Pipeline, synthetic_raw, v180_benchmark_params
N, A, B = 1024, 12, 2
raw = synthetic_raw(N, A, B, seed=4)
p = v180_benchmark_params(N, A, B)
p.fixedPatternNoiseRemoval = 0
o = common.make_oracle(p); want = o.process(raw)
pipe = Pipeline(p, device=0)
d = torch.from_numpy(raw.view(np.int16)).to('cuda:0')
pipe.process_device(d.data_ptr()); pipe.synchronize(); g1 = pipe.processed_host()
pipe.debug_force_prepared(True)
pipe.process_device(d.data_ptr()); pipe.synchronize(); g2 = pipe.processed_host()
print('u16 vs oracle', np.abs(g1-want).max())
print('f32 vs oracle', np.abs(g2-want).max())
e = np.abs(g2-want).reshape(A*B, N//2)
print('per line max err', e.max(axis=1))
s1 = pipe.debug_spectrum(d.data_ptr(), A*B).reshape(A*B, N)
os_ = o.last_spectrum().reshape(A*B, N)
print('spec err per line', np.abs(s1-os_).max(axis=1)/np.abs(os_).max(axis=1))
