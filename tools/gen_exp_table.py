#!/usr/bin/env python3
"""Constants of the spec's exp_: T[j] = 2^(j/64) correctly rounded to binary64, ln2/64 split in
two doubles, 64/ln2.  Exact decimal arithmetic (60 digits); prints C initialisers as hex floats."""
from decimal import Decimal, getcontext

getcontext().prec = 60
LN2 = Decimal("0.693147180559945309417232121458176568075500134360255254120680")
L = LN2 / 64
l_hi = float(L)
l_lo = float(L - Decimal(l_hi))
inv = float(64 / LN2)
print("inv_l  = %s  /* %r */" % (inv.hex(), inv))
print("l_hi   = %s  /* %r */" % (l_hi.hex(), l_hi))
print("l_lo   = %s  /* %r */" % (l_lo.hex(), l_lo))
tab = [float((LN2 * j / 64).exp()) for j in range(64)]
rows = []
for i in range(0, 64, 4):
    rows.append("    " + ", ".join(t.hex() for t in tab[i:i + 4]) + ",")
print("\n".join(rows))
