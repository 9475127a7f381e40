import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, torch, common
from octproz_amd import Pipeline, synthetic_raw, v180_benchmark_params
N, A, B = 1024, 24, 2
raw = synthetic_raw(N, A, B, seed=4)
p = v180_benchmark_params(N, A, B)
o = common.make_oracle(p); want = o.process(raw)
pipe = Pipeline(p, device=0)
pipe.set_mean_line(o.mean_line(), pin=True)
d = torch.from_numpy(raw.view(np.int16)).to('cuda:0')
pipe.process_device(d.data_ptr()); pipe.synchronize(); g1 = pipe.processed_host()
m1 = pipe.mean_line()
pipe.debug_force_prepared(True)
pipe.process_device(d.data_ptr()); pipe.synchronize(); g2 = pipe.processed_host()
m2 = pipe.mean_line()
print('mean same', np.array_equal(m1, m2), np.abs(m1).max())
print('u16 vs oracle', np.nanmax(np.abs(g1-want)))
print('f32 vs oracle', np.nanmax(np.abs(g2-want)))
print('g1==g2', np.array_equal(g1, g2))
pipe.debug_force_prepared(False)
pipe.process_device(d.data_ptr()); pipe.synchronize(); g3 = pipe.processed_host()
print('g1==g3', np.array_equal(g1, g3), 'g2==g3', np.array_equal(g2,g3))
print(g1[:4], g2[:4], g3[:4], want[:4])
