// Boundary I/O of the HIP pairing engine: the reference's #[repr(C)] images <-> engine values.
//   G1 = 24 u32 (x,y,z Fq)   G2 = 48 u32 (x,y,z Fq2)   Gt/Fq12 = 96 u32 in the order c0.c0.c0, c0.c0.c1, c0.c1.c0, ... c1.c2.c1
#pragma once
#include "pairing.hpp"

namespace bn254 {

#define F2P ((const F2 *)nullptr)
template <class F2>
BN_FN Fq12<F2> f12_load(const uint32_t *w) {
    Fq12<F2> f;
    f.c0.c0 = f2_load(F2P, w + 0);  f.c0.c1 = f2_load(F2P, w + 16); f.c0.c2 = f2_load(F2P, w + 32);
    f.c1.c0 = f2_load(F2P, w + 48); f.c1.c1 = f2_load(F2P, w + 64); f.c1.c2 = f2_load(F2P, w + 80);
    return f;
}
template <class F2>
BN_FN void f12_store(const Fq12<F2> &f, uint32_t *w) {
    f2_store(f.c0.c0, w + 0);  f2_store(f.c0.c1, w + 16); f2_store(f.c0.c2, w + 32);
    f2_store(f.c1.c0, w + 48); f2_store(f.c1.c1, w + 64); f2_store(f.c1.c2, w + 80);
}
BN_FN bool words_all_zero(const uint32_t *w, int n) {
    uint32_t o = 0;
    for (int i = 0; i < n; ++i) o |= w[i];
    return o == 0;
}
#undef F2P
}  // namespace bn254
