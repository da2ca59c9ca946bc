"""Replays the order in which a wave of moe_gemm_fp8w_s128.hip issues its vector-memory operations (LDS-DMA pieces of the
activation ring X, weight-fragment loads A) and prints, per stage, how many operations are younger than the last one the stage's
sync point needs (X(t+1) and A(t+1)): that is the literal of its `s_waitcnt vmcnt(N)`.  Any smaller literal is safe, a larger one
is a race.  The kernel uses 9 in the steady state (stage 0 could take 12), 5 for stage T-3, 0 for stage T-2.

    python tools/s128_waits.py [weight stages ahead = 2]"""
import sys


def replay(T, adepth):
    ops = []
    X = lambda t, p: ops.append(("X", t, p))
    A = lambda t, r, k: ops.append(("A", t, r, k))
    for st in range(2):                       # prologue: X0 A0 X1 A1 ...
        for p in range(4):
            X(st, p)
        for r in range(2):
            for k in range(2):
                A(st, r, k)
    if adepth == 3:
        for r in range(2):
            for k in range(2):
                A(2, r, k)
    for p in range(4):                        # ... X2, half of X3
        X(2, p)
    X(3, 0)
    X(3, 1)
    res = {}
    for t in range(T):
        lda, carry, dmax = t + adepth < T, t + 3 < T, t + 4 < T
        if lda:
            A(t + adepth, 0, 0), A(t + adepth, 0, 1)          # slot 0
        if carry:
            X(t + 3, 2)                                       # slot 2
        if lda:
            A(t + adepth, 1, 0), A(t + adepth, 1, 1)          # slot 4
        if carry:
            X(t + 3, 3)                                       # slot 6
        if t + 1 < T:                                         # sync point behind slot 13
            need = [i for i, o in enumerate(ops) if o[1] == t + 1]
            res[t] = len(ops) - 1 - max(need)
        if dmax:
            X(t + 4, 0), X(t + 4, 1)                          # slots 14, 15
    return res


if __name__ == "__main__":
    adepth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    for T in (4, 6, 8, 12, 32):
        r = replay(T, adepth)
        print(f"T = {T:2d} stages, weights {adepth} ahead: vmcnt per stage {[r[t] for t in sorted(r)]}")
