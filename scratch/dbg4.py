import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, torch, common
from octproz_amd import Pipeline, synthetic_raw, v180_benchmark_params
N, A, B = 1024, 8, 2
raw = synthetic_raw(N, A, B, seed=4)
p = v180_benchmark_params(N, A, B)
p.fixedPatternNoiseRemoval = 0; p.resampling = 0; p.windowing = 0; p.dispersionCompensation = 0
pipe = Pipeline(p, device=0)
d = torch.from_numpy(raw.view(np.int16)).to('cuda:0')
pipe.debug_force_prepared(True)
s2 = pipe.debug_spectrum(d.data_ptr(), A*B).reshape(A*B, N)
x = np.fft.fft(s2.astype(np.complex128), axis=1) / N   # inverse of the unnormalised inverse DFT
x = x.real
r = raw.reshape(A*B, N).astype(np.float64)
bad = np.abs(x - r) > 0.5
print('bad count per line', bad.sum(axis=1))
l = 0
idx = np.nonzero(bad[l])[0]
print('line0 bad idx', idx[:40], '...', idx[-10:] if len(idx) else '')
if len(idx):
    print('got', np.round(x[l, idx[:8]], 1), 'want', r[l, idx[:8]])
    for j in idx[:4]:
        m = np.argmin(np.abs(r[l] - x[l, j])); print(j, '->', m, r[l, m], x[l, j])
