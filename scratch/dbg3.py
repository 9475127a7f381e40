import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, torch, common
from octproz_amd import Pipeline, synthetic_raw, v180_benchmark_params
N, A, B = 1024, 24, 2
raw = synthetic_raw(N, A, B, seed=4)
p = v180_benchmark_params(N, A, B)
p.fixedPatternNoiseRemoval = 0
o = common.make_oracle(p); want = o.process(raw)
os_ = o.last_spectrum().reshape(A*B, N)
pipe = Pipeline(p, device=0)
d = torch.from_numpy(raw.view(np.int16)).to('cuda:0')
s1 = pipe.debug_spectrum(d.data_ptr(), A*B).reshape(A*B, N)
print('u16 spec err', (np.abs(s1-os_).max(axis=1)/np.abs(os_).max(axis=1)).max())
pipe.debug_force_prepared(True)
s2 = pipe.debug_spectrum(d.data_ptr(), A*B).reshape(A*B, N)
e = np.abs(s2-os_).max(axis=1)/np.abs(os_).max(axis=1)
print('f32 spec err per line', e)
u = pipe.debug_unpack(d.data_ptr(), N*A*B)
print('unpack ok', np.array_equal(u, raw.reshape(-1).astype(np.float32)))
pipe.process_device(d.data_ptr()); pipe.synchronize(); g2 = pipe.processed_host()
print('f32 img err', np.abs(g2-want).max())
