import numpy as np
from scipy.special import erf
from scipy.optimize import least_squares
np.set_printoptions(precision=17)
X = 5.5
def target(x): return 0.5*erf(x/np.sqrt(2.0))   # Phi(x) - 0.5, odd
# R(x) = x P(t)/Q(t), t=x^2; P deg np_, Q deg nq (q0 = 1)
def fit(npd, nqd, X, iters=60):
    # Chebyshev nodes in x on (0, X]
    n = 4000
    x = X*np.cos(np.pi*(np.arange(n)+0.5)/(2*n))   # in (0,X)
    t = x*x
    f = target(x)
    w = np.ones_like(x)
    Qv = np.ones_like(x)
    best=None
    for it in range(iters):
        # minimize w*(x P - f Q)/Qv ; unknowns p0..pnp, q1..qnq
        A = np.concatenate([ (x[:,None]*t[:,None]**np.arange(npd+1)), -(f[:,None]*t[:,None]**np.arange(1,nqd+1)) ],axis=1)
        b = f
        s = (w/Qv)[:,None]
        # column scaling
        cs = np.abs(A).max(axis=0)
        sol,*_ = np.linalg.lstsq(A*s/cs, b*s[:,0], rcond=None)
        sol = sol/cs
        p = sol[:npd+1]; q = np.concatenate([[1.0], sol[npd+1:]])
        Pv = sum(p[k]*t**k for k in range(npd+1)); Qv = sum(q[k]*t**k for k in range(nqd+1))
        err = x*Pv/Qv - f
        # error on gelu = x * err (use weight so that |err|*max(1,|x|) is equalised): Lawson reweighting
        e = np.abs(err)*np.maximum(1.0, x)
        if best is None or e.max() < best[0]: best=(e.max(), p.copy(), q.copy())
        w = w*(0.5+ e/e.max())**1.0 * np.maximum(1.0,x)**0  # Lawson
        w = w/w.max()
    return best
for npd,nqd in [(4,4),(5,4),(5,5),(6,4),(6,5),(6,6)]:
    e,p,q = fit(npd,nqd,X)
    print(npd,nqd,"max weighted err %.3e"%e, "Q min", min(sum(q[k]*tt**k for k in range(nqd+1)) for tt in np.linspace(0,X*X,1000)))

def eval32(p,q,x,X):
    f=np.float32
    x=x.astype(f); xc=np.clip(x,f(-X),f(X)); t=xc*xc
    P=np.full_like(x,f(p[-1]))
    for c in p[-2::-1]: P=(P*t+f(c)).astype(f)
    Q=np.full_like(x,f(q[-1]))
    for c in q[-2::-1]: Q=(Q*t+f(c)).astype(f)
    r=(xc*P).astype(f)*(f(1)/Q).astype(f)
    return (x*(f(0.5)+r)).astype(f)   # fma(x, r, 0.5x) alternative
xs=np.concatenate([np.linspace(-12,12,2000001), np.random.default_rng(0).standard_normal(2000000)*2])
truth=xs.astype(np.float32).astype(np.float64); truth=0.5*truth*(1+erf(truth/np.sqrt(2)))
ref32=(np.float32(0.5)*xs.astype(np.float32)*(np.float32(1)+erf((xs.astype(np.float32)*np.float32(0.7071067811865476)).astype(np.float32)).astype(np.float32))).astype(np.float32)
print("fp32 reference form: max abs err %.3e  max err/ max(1,|x|) %.3e"%(np.abs(ref32-truth).max(), (np.abs(ref32-truth)/np.maximum(1,np.abs(xs))).max()))
for npd,nqd in [(5,5),(6,4),(5,4)]:
    e,p,q=fit(npd,nqd,X)
    g=eval32(p,q,xs,X)
    d=np.abs(g-truth)
    print(npd,nqd,"fp32 eval: max abs err %.3e  rel-to-max(1,|x|) %.3e"%(d.max(), (d/np.maximum(1,np.abs(xs))).max()), "at x=",xs[d.argmax()])
    print(" p=",[float(v) for v in p]); print(" q=",[float(v) for v in q])
