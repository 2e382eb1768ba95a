#!/usr/bin/env python3
"""Constants of the spec's exponentials (bmm-mcmc_amd/csrc/bmm_spec.h), as C hex floats.

  python tools/gen_exp_table.py 64    exp_  : T[j] = 2^(j/64),  ln2/64 in two doubles, 64/ln2
  python tools/gen_exp_table.py 256   expw_ : T[j] = 2^(j/256), ln2/256 in two doubles, 256/ln2

Exact decimal arithmetic (60 digits); float(Decimal) rounds correctly to binary64."""
import sys
from decimal import Decimal, getcontext

getcontext().prec = 60
LN2 = Decimal("0.693147180559945309417232121458176568075500134360255254120680")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = LN2 / n
l_hi = float(L)
l_lo = float(L - Decimal(l_hi))
inv = float(n / LN2)
print("inv_l  = %s  /* %r */" % (inv.hex(), inv))
print("l_hi   = %s  /* %r */" % (l_hi.hex(), l_hi))
print("l_lo   = %s  /* %r */" % (l_lo.hex(), l_lo))
tab = [float((LN2 * j / n).exp()) for j in range(n)]
for i in range(0, n, 4):
    print("    " + ", ".join(t.hex() for t in tab[i:i + 4]) + ", \\")
