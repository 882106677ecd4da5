#!/usr/bin/env python3
"""Minimax polynomials for kid_amd/csrc/fastmath.h (run once; the coefficients printed here are pasted into the header).

  2**r = 1 + r*P2(r)   on |r| <= 0.52   (exp2_small: the remainder of an exponent rounded to the nearest integer)
  e**r = 1 + r*Pe(r)   on |r| <= 0.36   (exp_small: |r| <= ln2/2 after the reduction by n*ln2)

Remez exchange in 60-digit arithmetic (mpmath) for the RELATIVE error of the function value, then the coefficients are
rounded to binary64 and the error of the rounded polynomial (evaluated exactly) is measured on a dense grid: that figure, not
the ideal one, is what the header quotes.  Degree 10 for P (11 for the function) leaves < 1e-17, a fifth of half an ulp;
the degree-13 Taylor polynomials they replace needed three more coefficients for the same error."""
import sys

import mpmath as mp

mp.mp.dps = 60


def remez(g, weight, a, n, iters=30, avoid_zero=False):
    """minimax p (degree n) of g on [-a, a] for the weighted error weight(x)*(p(x) - g(x))"""
    xs = [a * mp.cos(mp.pi * (2 * i + 1) / (2 * (n + 2))) for i in range(n + 2)][::-1]          # n + 2 Chebyshev points
    xs = [x if abs(x) > a / 1000 else a / 50 for x in xs]                                          # (the weight vanishes at 0: no extremum there)
    for _ in range(iters):
        A = mp.matrix(n + 2, n + 2)
        b = mp.matrix(n + 2, 1)
        for i, x in enumerate(xs):
            for j in range(n + 1):
                A[i, j] = x ** j
            A[i, n + 1] = (-1) ** i / weight(x)
            b[i] = g(x)
        sol = mp.lu_solve(A, b)
        c = [sol[j] for j in range(n + 1)]
        err = lambda x: weight(x) * (mp.polyval(c[::-1], x) - g(x))
        # new extrema: the largest |err| between consecutive sign changes on a dense grid, polished by golden section
        grid = [-a + 2 * a * mp.mpf(i) / 4000 for i in range(4001)]
        if avoid_zero:
            grid = [x for x in grid if abs(x) > a / 3000]
        vals = [err(x) for x in grid]
        ext, cur = [], 0
        for i in range(1, len(grid)):
            if mp.sign(vals[i]) != mp.sign(vals[cur]) and vals[i] != 0:
                seg = range(cur, i)
                k = max(seg, key=lambda q: abs(vals[q]))
                ext.append(grid[k])
                cur = i
        k = max(range(cur, len(grid)), key=lambda q: abs(vals[q]))
        ext.append(grid[k])
        if len(ext) != n + 2:
            break
        if max(abs(e1 - e0) for e0, e1 in zip(xs, ext)) < a * mp.mpf(10) ** -12:
            xs = ext
            break
        xs = ext
    return c, abs(sol[n + 1])


def measure(cd, f, a):
    worst = mp.mpf(0)
    for i in range(20001):
        x = -a + 2 * a * i / 20000
        p = mp.mpf(0)
        for q in reversed(cd):
            p = p * x + mp.mpf(q)
        worst = max(worst, abs((1 + x * p) / f(x) - 1))
    return worst


def report(name, f, a, n, taylor):
    a = mp.mpf(a)
    g = lambda x: (f(x) - 1) / x if x != 0 else mp.diff(f, 0)
    w = lambda x: (abs(x) if x != 0 else mp.mpf(10) ** -40) / f(x)                 # relative error of 1 + x p(x)
    c, e = remez(g, w, a, n)
    # the two lowest coefficients are rounded to binary64 one at a time and the rest re-fitted after each rounding, so that
    # the rounding error of ln 2 (or of p1 ~ 1/2) is absorbed by the higher coefficients as far as it can be
    cd = []
    for k in (1, 2):
        cd.append(float(c[0]))
        head = lambda x, cd=tuple(cd): sum(mp.mpf(q) * x ** i for i, q in enumerate(cd))
        gk = lambda x, k=k, head=head: ((f(x) - 1) / x - head(x)) / x ** k
        wk = lambda x, k=k: abs(x) ** (k + 1) / f(x)
        c, _ = remez(gk, wk, a, n - k, avoid_zero=True)
    cd += [float(x) for x in c]
    print(f"// {name}: 1 + r*P(r), P of degree {n}, |r| <= {mp.nstr(a, 3)}: ideal relative error {mp.nstr(e, 3)}, with binary64 "
          f"coefficients {mp.nstr(measure(cd, f, a), 3)} (the degree-12 Taylor P it replaces, binary64 coefficients: {mp.nstr(measure(taylor, f, a), 3)})")
    print("   ", ", ".join(f"p{i} = {x!r}" for i, x in enumerate(cd)))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    report("exp2_small", lambda x: mp.mpf(2) ** x, "0.52", n, [float(mp.log(2) ** k / mp.factorial(k)) for k in range(1, 14)])
    report("exp_small", lambda x: mp.e ** x, "0.36", n, [float(1 / mp.factorial(k)) for k in range(1, 14)])
