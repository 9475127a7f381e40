import os, sys
base = open('kernels.h').read()
SUBS = {
 'nofft': ("		fft_wave<LOG2N, !SPECTRUM>(v, xbuf, tw, lane);", "#pragma unroll\n		for (int q = 0; q < P; q++) asm volatile(\"\" : \"+v\"(v[q]));"),
 'nogather': ("""				const int n1 = (int)L.x;
				const float* t = &row[ROW_OFF - 1 + n1];
				y = cubic_hermite(t[0], t[1], t[2], t[3], L.x - (float)n1);""", "				y = rowl[64 * q] * L.x;"),
 'nolut': ("			if constexpr (LDS_LUT) { const float4 t = lutL[lane + 64 * q]; L = f32x4{t.x, t.y, t.z, t.w}; }", "			if constexpr (LDS_LUT) { L = f32x4{1.5f, 1.0f, 1.0f, 0.5f}; }"),
 'nostore': ("				if constexpr (NBL == 4) buf_store128(f32x4{o[0], o[1], o[2], o[3]}, outR, lane * 16, u * (N / RL) * 4);", "				if constexpr (NBL == 4) { if (o[0] == 1234.5f) buf_store128(f32x4{o[0], o[1], o[2], o[3]}, outR, lane * 16, u * (N / RL) * 4); }"),
 'nostage': ("					*reinterpret_cast<float4*>(&row[ROW_OFF + SPL * lane + 64 * SPL * i + 4 * h]) = chunk_to_float<INTYPE>(pre[i], h, shift);", "					{ float4 f_ = chunk_to_float<INTYPE>(pre[i], h, shift); if (f_.x == 1234.5f) *reinterpret_cast<float4*>(&row[ROW_OFF + SPL * lane + 64 * SPL * i + 4 * h]) = f_; }"),
 'noepi': ("					const float s = LOGSCALE ? __builtin_amdgcn_logf(p) : __builtin_amdgcn_sqrtf(p);", "					const float s = p;"),
}
def mk(name, keys):
    s = base
    for k in keys:
        a, b = SUBS[k]
        assert a in s, k
        s = s.replace(a, b)
    d = '/tmp/abl_' + name
    os.makedirs(d, exist_ok=True)
    for f in ('launch.h', 'fused_inst.hip', 'fft_regs.h'):
        open(d + '/' + f, 'w').write(open(f).read())
    open(d + '/kernels.h', 'w').write(s)
for spec in sys.argv[1:]:
    name, keys = spec.split('=')
    mk(name, keys.split('+'))
