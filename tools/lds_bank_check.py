#!/usr/bin/env python3
"""Bank-conflict enumeration of the weight-gradient kernel's LDS image (csrc/ffx.hip `WgT`), by the lane groups and bank moduli of
/opt/skills/guides/MI355X_MICROARCH.md ("LDS access groups"): for every read / write instruction of the tile loop, the number of
LDS cycles its lane groups need beyond the conflict-free count.  Run: python tools/lds_bank_check.py"""
import itertools

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]
HALVES = [list(range(32)), list(range(32, 64))]
W64_GROUPS = [list(range(16 * k, 16 * k + 16)) for k in range(4)]


def cycles(groups, addr, nbytes, modulus):
    """extra cycles: per group, max over banks of (distinct 4-byte words on that bank) - 1"""
    extra = 0
    for g in groups:
        banks = {}
        for l in g:
            for b in range(0, nbytes, 4):
                a = addr(l) + b
                banks.setdefault((a // 4) % modulus, set()).add(a // 4)
        extra += max(len(v) for v in banks.values()) - 1
    return extra


def check(C):
    prow, wrap = (192, 0) if C == 64 else (64, 64)
    row = lambda R: R * prow + 16 * ((R >> 2) & 3) + wrap * (R >> 4)      # noqa: E731
    plane = 32 * prow + 48 + wrap + 16
    # rows must not overlap
    spans = sorted((row(R), row(R) + 2 * C) for R in range(32))
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] <= plane, (C, spans)
    worst = {}
    for q in range(C // 16):       # pixel-major ds_read_b128: lane (j, half) reads row j at 32 q + 16 half
        e = cycles(B128_GROUPS, lambda l: row(l & 31) + 32 * q + 16 * (l >> 5), 16, 64)
        worst["ds_read_b128 pixel-major"] = max(worst.get("ds_read_b128 pixel-major", 0), e)
    for mt, s2, r2 in itertools.product(range(C // 32), range(2), range(2)):
        def addr(l):
            half, g, i = l >> 5, (l >> 4) & 1, l & 15
            return ((4 * half + (i >> 2)) * prow + 16 * half + 32 * g + 8 * (i & 3) + s2 * (16 * prow + wrap)
                    + r2 * (8 * prow + 32) + 64 * mt)
        # (the address of lane l must be row(16 s2 + 8 r2 + 4 half + jj) + 64 mt + 32 g + 8 q')
        for l in range(64):
            half, g, i = l >> 5, (l >> 4) & 1, l & 15
            assert addr(l) == row(16 * s2 + 8 * r2 + 4 * half + (i >> 2)) + 64 * mt + 32 * g + 8 * (i & 3)
        e = cycles(HALVES, addr, 8, 64)
        worst["ds_read_b64_tr_b16"] = max(worst.get("ds_read_b64_tr_b16", 0), e)
    nt = 512 if C == 64 else 256
    for w in range(nt // 64):      # staging ds_write_b64: thread f writes 8 bytes of row f / (C/4) at 8 (f % (C/4))
        e = cycles(W64_GROUPS, lambda l: row((64 * w + l) // (C // 4)) + 8 * ((64 * w + l) % (C // 4)), 8, 32)
        worst["ds_write_b64 staging"] = max(worst.get("ds_write_b64 staging", 0), e)
    print(f"C = {C}: row stride {prow} B, plane {plane} B, extra LDS cycles per instruction (0 = conflict-free): {worst}")
    return worst


if __name__ == "__main__":
    for C in (64, 32):
        check(C)
