"""Fits the polynomial of bert.hip::erf_poly ((log erfc(t) + t) / t on [0, 4], Lawson-weighted least squares in float64 against
40-digit mpmath values) and measures the resulting GELU in fp32 arithmetic on [-8, 8]."""
import numpy as np, mpmath as mp
mp.mp.dps = 40
def lawson(f, lo, hi, deg, n=3000, iters=300):
    k = np.arange(n); x = 0.5*(lo+hi) + 0.5*(hi-lo)*np.cos(np.pi*(k+0.5)/n)
    y = np.array([float(f(mp.mpf(float(v)))) for v in x])
    w = np.ones(n)
    for it in range(iters):
        c = np.polynomial.polynomial.polyfit(x, y, deg, w=np.sqrt(w))
        e = np.abs(np.polynomial.polynomial.polyval(x, c) - y)
        w = w * e; w /= w.sum(); w += 1e-12
    return c, e.max()
h = lambda t: (mp.log(mp.erfc(t)) + t)/t if t > 0 else mp.mpf(1) - 2/mp.sqrt(mp.pi)
for deg in (7, 8, 9):
    c, e = lawson(h, 1e-9, 4.0, deg)
    f32 = np.float32
    c32 = c.astype(f32)
    xs = np.linspace(-8, 8, 400001).astype(f32)
    a = (xs * f32(0.70710678)).astype(f32)
    t = np.minimum(np.abs(a), f32(4.0))
    q = np.full_like(t, c32[-1])
    for cc in c32[-2::-1]: q = (q*t + cc).astype(f32)
    arg = (q*t - t).astype(f32)
    erf = np.copysign((f32(1.0) - np.exp(arg.astype(np.float64)).astype(f32)).astype(f32), a)
    gelu = (f32(0.5)*xs*(f32(1.0)+erf)).astype(f32)
    ref = np.array([float(mp.mpf(float(v))/2*(1+mp.erf(mp.mpf(float(v))/mp.sqrt(2)))) for v in xs[::40]])
    err = np.abs(gelu[::40].astype(np.float64) - ref)
    rel = err / np.maximum(np.abs(ref), 1e-30)
    print(deg, "fit err", e, "gelu max abs err", err.max(), "at", xs[::40][err.argmax()], "max abs err / max(|x|,1)", (err/np.maximum(np.abs(xs[::40]),1)).max())
c, e = lawson(h, 1e-9, 4.0, 8)
print(['%.9ef' % float(np.float32(v)) for v in c])
