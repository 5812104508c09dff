#!/usr/bin/env python3
"""How many operand bytes must the eight L2s of an MI355X fetch for one round of the persistent 256 x 256 GEMM -- for the product's workgroup -> tile map and for the
alternatives?  (CPU only; VERDICT r4 item 3: "reads are 3.0 x algorithmic because each of 8 XCDs streams its own A / B panels".)

The dispatcher hands workgroup b to XCD b % 8; `xcd_remap` (csrc/common.hip.h) gives each XCD a CONTIGUOUS run of G / 8 logical ids, and the grouped raster
(`raster_decode`: groups of 4 tile rows, walked column by column) turns a run of 32 ids into a block of 4 tile rows x 8 tile columns.  An XCD's L2 then fetches, per round,
the A panels of its distinct tile rows and the B panels of its distinct tile columns ONCE each (every CU of the XCD walks K in step).  This tool counts them.

    python tools/xcd_panel_traffic.py [M N K ...]          # default: the headline 4096^3, C3, 8192^3
The measured FETCH_SIZE of the 4096^3 launch is 52.6 MB (profiles/rocprof_pmc_gemm_r5a_mxfp4_4096.txt, x 2 as the guide prescribes; round 4: 53.8 MB): the model below
gives 51.1 MB for the product map -- every panel is fetched once per XCD and no more; what is left of the 3.0 x is the price of eight separate L2s."""
import math
import sys


def xcd_remap(b, nb):
    q, r, xcd, idx = nb >> 3, nb & 7, b & 7, b >> 3
    return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + idx


def raster(t, tiles_m, tiles_n, gm=4):
    group = gm * tiles_n
    gid, rem = divmod(t, group)
    first = gid * gm
    gsz = min(tiles_m - first, gm)
    return first + rem % gsz, rem // gsz


def traffic(M, N, K, mapping, cus=256, ebits=4):
    tm, tn = -(-M // 256), -(-N // 256)
    T = tm * tn
    rounds = -(-T // cus)
    G = min(cus, (-(-T // rounds) + 7) // 8 * 8, T)
    panel = 256 * K * ebits // 8 * (1 + 1 / 16)            # operand bytes of one tile row / column + its e8m0 scales (1 byte per 32 elements = 1/16 at 4 bit)
    total = 0
    for rnd in range(rounds):
        rows = [set() for _ in range(8)]
        cols = [set() for _ in range(8)]
        for b in range(G):
            t = mapping(b, G, rnd, tm, tn)
            if t is None or t >= T:
                continue
            r, c = raster(t, tm, tn) if mapping is not plain_rowmajor else divmod(t, tn)
            rows[b % 8].add(r); cols[b % 8].add(c)
        total += sum(len(rows[x]) + len(cols[x]) for x in range(8)) * panel
    return total, (tm + tn) * panel, G, rounds


def product(b, G, rnd, tm, tn):       # gemm_mx_deepp: tile = xcd_remap(b, G) + rnd * G, grouped raster of 4 tile rows
    return xcd_remap(b, G) + rnd * G


def no_remap(b, G, rnd, tm, tn):      # what the dispatcher's round-robin gives without xcd_remap: an XCD's tiles are 8 apart in the raster
    return b + rnd * G


def plain_rowmajor(b, G, rnd, tm, tn):   # xcd_remap + plain row-major tile order (no grouped raster)
    return xcd_remap(b, G) + rnd * G


def main():
    shapes = [tuple(int(v) for v in sys.argv[i:i + 3]) for i in range(1, len(sys.argv) - 2, 3)] or [(4096, 4096, 4096), (4096, 14336, 4096), (8192, 8192, 8192), (5120, 4096, 4096)]
    print(f"{'M x N x K':22s} {'G':>4s} {'rounds':>6s} {'algorithmic':>12s} | {'product map':>12s} {'x alg':>6s} | {'no xcd_remap':>12s} {'x alg':>6s} | {'row-major':>10s} {'x alg':>6s} | square-block bound")
    for (M, N, K) in shapes:
        res = {}
        for name, mp in (("product", product), ("noremap", no_remap), ("rowmajor", plain_rowmajor)):
            res[name] = traffic(M, N, K, mp)
        alg, G, rounds = res["product"][1], res["product"][2], res["product"][3]
        per_xcd = G / 8
        panel = 256 * K // 2 * (1 + 1 / 16)
        bound = rounds * 8 * 2 * math.sqrt(per_xcd) * panel      # a square block of G / 8 tiles touches 2 sqrt(G / 8) panels: no map of whole tiles can do better
        p, n, r = res["product"][0], res["noremap"][0], res["rowmajor"][0]
        print(f"{M} x {N} x {K:<8d} {G:4d} {rounds:6d} {alg / 1e6:10.1f} MB | {p / 1e6:10.1f} MB {p / alg:6.2f} | {n / 1e6:10.1f} MB {n / alg:6.2f} | {r / 1e6:8.1f} MB {r / alg:6.2f} | {bound / 1e6:8.1f} MB ({p / bound:.2f} x)")


if __name__ == "__main__":
    main()
