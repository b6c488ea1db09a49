// Round 5's wider version of tools/exp_chain_search.py (which covers up to three extra digits below 130 in ~30 minutes of Python): the fewest non-zero
// signed digits of the BN parameter u over every digit set {1, d1, d2} of odd digits below a bound (dynamic programming over bit position and carry),
// priced with a table-building model (each digit one product a*2^j +- b from what is known) and a dense product = 2.72 cyclotomic squarings (7147 /
// 2632 instructions in the lane-pair mapping).  gcc -O2 tools/exp_chain_search.c -o /tmp/search && /tmp/search 2048   (about 40 minutes):
// +-17, +-35 (12 non-zero digits, 2 products + 5 squarings of table: 62 squarings + 13 products per exponentiation) stays the optimum for all pairs
// of digits below 2048 - the chain of pairing.hpp / gen_device_constants.py is not improvable by a different digit pair.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
typedef unsigned __int128 u128;
static const uint64_t U = 4965661367192848881ull;
#define LEV 80
#define C 2100
static int cost[LEV + 1][2 * C + 1];
static int recode(const int *D, int nd, int *top_out) {
    // k_i = (U >> i) + c
    for (int i = LEV; i >= 0; --i)
        for (int c = -C; c <= C; ++c) {
            __int128 k = (i < 64 ? (__int128)(U >> i) : 0) + c;
            int best = 1 << 20;
            if (k == 0) best = 0;
            else if (k > 0 && i < LEV) {
                if ((k & 1) == 0) {
                    __int128 k2 = k >> 1; __int128 c2 = k2 - (i + 1 < 64 ? (__int128)(U >> (i + 1)) : 0);
                    if (c2 >= -C && c2 <= C) best = cost[i + 1][(int)c2 + C];
                } else {
                    for (int j = 0; j < nd; ++j) for (int sg = -1; sg <= 1; sg += 2) {
                        __int128 r = k - sg * D[j];
                        if (r < 0) continue;
                        __int128 k2 = r >> 1; __int128 c2 = k2 - (i + 1 < 64 ? (__int128)(U >> (i + 1)) : 0);
                        if (c2 < -C || c2 > C) continue;
                        int v = cost[i + 1][(int)c2 + C] + 1;
                        if (v < best) best = v;
                    }
                }
            }
            cost[i][c + C] = best;
        }
    (void)top_out;
    return cost[0][C];
}
// cheapest way to build digit d from known set K by ONE product: d = a*2^j +- b (a, b in K): returns squarings j, or -1
static int build1(int d, const int *K, int nk) {
    int bestj = -1;
    for (int x = 0; x < nk; ++x) for (int y = 0; y < nk; ++y) for (int j = 0; j < 12; ++j) {
        long a = K[x], b = K[y];
        if (a * (1L << j) + b == d || a * (1L << j) - b == d || b - a * (1L << j) == d) { if (bestj < 0 || j < bestj) bestj = j; }
    }
    return bestj;
}
int main(int argc, char **argv) {
    int maxd = argc > 1 ? atoi(argv[1]) : 1024;
    double M = 2.72;           // a dense product in cyclotomic squarings (7147 / 2632 instructions)
    double bestc = 1e9;
    for (int d1 = 3; d1 < maxd; d1 += 2) {
        for (int d2 = d1; d2 < maxd; d2 += 2) {          // d2 == d1: a single extra digit
            int D[3] = {1, d1, d2}; int nd = d2 == d1 ? 2 : 3;
            int nz = recode(D, nd, 0);
            if (nz >= (1 << 20)) continue;
            // table: try both orders
            int K[3] = {1, 0, 0}; int muls = 0, sq = 0, ok = 1;
            int order[2][2] = {{d1, d2}, {d2, d1}}; int bestm = 99, bests = 99;
            for (int o = 0; o < (nd == 3 ? 2 : 1); ++o) {
                K[0] = 1; int nk = 1; muls = 0; sq = 0; ok = 1;
                for (int t = 0; t < nd - 1; ++t) { int j = build1(order[o][t], K, nk); if (j < 0) { ok = 0; break; } K[nk++] = order[o][t]; muls++; sq += j; }
                if (ok && muls * M + sq < bestm * M + bests) { bestm = muls; bests = sq; }
            }
            if (bestm == 99) continue;
            // squarings of the main chain: the top non-zero position (approx. 62 - log2(top digit)); use 63 - bits(top digit) as an estimate
            double total = (nz - 1 + bestm) * M + bests;        // + main squarings (~ the same for all sets: compared without)
            if (total < bestc + 0.01) { bestc = total < bestc ? total : bestc; printf("digits 1 %d %d: nonzeros %d, table %d products + %d squarings, weighted %.2f\n", d1, d2, nz, bestm, bests, total); fflush(stdout); }
        }
    }
    return 0;
}
