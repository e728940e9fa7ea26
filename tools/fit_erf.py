import numpy as np, mpmath as mp
mp.mp.dps = 40
T = 0.927734375
def lawson(f, lo, hi, deg, n=3000, iters=200):
    k = np.arange(n); x = 0.5*(lo+hi) + 0.5*(hi-lo)*np.cos(np.pi*(k+0.5)/n)
    y = np.array([float(f(mp.mpf(float(v)))) for v in x])
    w = np.ones(n)
    for it in range(iters):
        c = np.polynomial.polynomial.polyfit(x, y, deg, w=np.sqrt(w))
        e = np.abs(np.polynomial.polynomial.polyval(x, c) - y)
        w = w * e; w /= w.sum(); w += 1e-12
    return c, e.max()
g = lambda s: (mp.erf(mp.sqrt(s))/mp.sqrt(s) - 1) if s > 0 else mp.mpf(2)/mp.sqrt(mp.pi) - 1
cs, es = lawson(g, 1e-12, T*T, 5)
print("small", es, [float(np.float32(c)) for c in cs])
h = lambda t: (mp.log(mp.erfc(t)) + t)/t
cl, el = lawson(h, T, 4.0, 6)
print("large", el, [float(np.float32(c)) for c in cl])
np.save('/tmp/cs.npy', cs); np.save('/tmp/cl.npy', cl)
# float32 evaluation check
f32 = np.float32
def erf32(a):
    a = a.astype(f32); t = np.abs(a); s = a*a
    cs32 = cs.astype(f32); cl32 = cl.astype(f32)
    r = np.full_like(a, cs32[5])
    for c in cs32[4::-1]: r = (r*s + c).astype(f32)      # not fused; fused is at least as good
    small = (r*a + a).astype(f32)
    tc = np.minimum(t, f32(4.0))
    q = np.full_like(a, cl32[6])
    for c in cl32[5::-1]: q = (q*tc + c).astype(f32)
    arg = (q*tc - tc).astype(f32)
    big = (f32(1.0) - np.exp(arg.astype(np.float64)).astype(f32)).astype(f32)
    big = np.copysign(big, a)
    return np.where(t > f32(T), big, small)
xs = np.concatenate([np.linspace(-6, 6, 2000001), np.random.default_rng(1).normal(size=1000000)*2]).astype(f32)
ref = np.array([float(mp.erf(mp.mpf(float(v)))) for v in xs[::50]])
got = erf32(xs[::50]).astype(np.float64)
ulp = np.spacing(np.abs(ref).astype(f32)).astype(np.float64)
err = np.abs(got-ref)/ulp
print("max ulp err", err.max(), "at", xs[::50][err.argmax()], "mean", err.mean())
