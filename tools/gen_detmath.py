#!/usr/bin/env python3
"""Print the constant tables of include/hk_detmath.h (computed with mpmath at 80 digits)."""
import mpmath as mp, struct
mp.mp.dps = 80

def d(x):           # nearest double
    return float(mp.mpf(x))

def trunc_bits(x, keep):
    """double with only the top `keep` mantissa bits of x (x>0)"""
    f = float(x); b = struct.unpack('<Q', struct.pack('<d', f))[0]
    b &= ~((1 << (52 - keep + 1)) - 1)
    return struct.unpack('<d', struct.pack('<Q', b))[0]

def lit(x):
    return "%.17g" % x

pio2 = mp.pi/2
p1 = trunc_bits(pio2, 33); r = pio2 - mp.mpf(p1)
p2 = trunc_bits(r, 33); r2 = r - mp.mpf(p2)
p3 = d(r2)
print("PIO2_1", lit(p1)); print("PIO2_2", lit(p2)); print("PIO2_3", lit(p3))
print("INV_PIO2", lit(d(2/mp.pi)))
print("PI_HI", lit(d(mp.pi)), "PI_LO", lit(d(mp.pi - mp.mpf(d(mp.pi)))))
print("PIO2_HI", lit(d(pio2)), "PIO2_LO", lit(d(pio2 - mp.mpf(d(pio2)))))
print("sin coeffs (r^3..r^21):", ", ".join(lit(d(mp.mpf((-1)**k)/mp.factorial(2*k+1))) for k in range(1, 11)))
print("cos coeffs (r^4..r^22):", ", ".join(lit(d(mp.mpf((-1)**k)/mp.factorial(2*k))) for k in range(2, 12)))
print("atan tab hi:", ", ".join(lit(d(mp.atan(mp.mpf(j)/8))) for j in range(9)))
print("atan tab lo:", ", ".join(lit(d(mp.atan(mp.mpf(j)/8) - mp.mpf(d(mp.atan(mp.mpf(j)/8))))) for j in range(9)))
print("atan coeffs (u^3..u^17):", ", ".join(lit(d(mp.mpf((-1)**k)/(2*k+1))) for k in range(1, 9)))
ln2 = mp.log(2)
l1 = trunc_bits(ln2, 32); print("LN2_HI", lit(l1), "LN2_LO", lit(d(ln2 - mp.mpf(l1))), "INV_LN2", lit(d(1/ln2)))
print("exp coeffs (r^2..r^14):", ", ".join(lit(d(1/mp.factorial(k))) for k in range(2, 15)))
