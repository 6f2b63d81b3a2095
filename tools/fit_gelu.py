"""Developer tool: derive and check the constants of lr_gelu_erf2 (leftrefill_amd/csrc/common.h).

Phi(-a) = 0.5 erfc(a / sqrt 2) is fitted as exp2(P7(a)) on a in [0, A]; the script prints the coefficients (highest
degree last) and the worst absolute GELU error of the float32 Horner evaluation the kernel performs."""
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as Pn
from scipy.special import erf, erfc

A = 4.3 * np.sqrt(2.0)
DEG = 7


def main():
    a = np.cos(np.pi * (np.arange(4000) + 0.5) / 4000) * A / 2 + A / 2
    f = np.log2(erfc(a / np.sqrt(2.0))) - 1.0
    coef = C.Chebyshev.fit(a, f, DEG, domain=[0, A]).convert(kind=Pn.Polynomial, domain=[0, A], window=[0, A]).coef
    print("A = %.9g" % A)
    print("coefficients c0..c%d:" % DEG, ", ".join("%.9e" % v for v in coef))
    x = np.linspace(-9, 9, 1800001).astype(np.float32)
    ac = np.minimum(np.abs(x), np.float32(A))
    p = np.float32(coef[-1]) * np.ones_like(ac)
    for k in range(DEG - 1, -1, -1):
        p = (p * ac + np.float32(coef[k])).astype(np.float32)
    e = np.exp2(p).astype(np.float32)
    d = np.copysign(np.float32(0.5) - e, x).astype(np.float32)
    y = (x * d + np.float32(0.5) * x).astype(np.float32)
    xr = x.astype(np.float64)
    ref = 0.5 * xr * (1.0 + erf(xr / np.sqrt(2.0)))
    err = np.abs(y - ref)
    print("max |GELU error| = %.3e at x = %.4f" % (err.max(), x[err.argmax()]))
    return err.max()


if __name__ == "__main__":
    main()
