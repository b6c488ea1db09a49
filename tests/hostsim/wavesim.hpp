// TEST INFRASTRUCTURE (host simulation only): executes the wave-cooperative Fq12 machine (bn_amd/csrc/wave.hpp) on the CPU.
// One OS thread per lane PAIR (32 per simulated wave) runs the very templates the kernels instantiate, with T = FeP (lanepair.hpp);
// the LDS register file is an array shared by the threads (limbs plus, under -DBN_BOUNDS, the tracked bounds of every slot), and
// the wave-level barrier of a phase is a pthread barrier.  Not linked into the product.
#pragma once
#include <pthread.h>
#include <vector>
#include "../../bn_amd/csrc/wave.hpp"

namespace bn254 { namespace wv {

struct WaveSimShared {
    uint32_t limb[NPAGES * PAGE_DW];
#if defined(BN_BOUNDS)
    uint32_t lb[NPAGES * PAGE_DW], vb[NPAGES * PAGE_DW];
    bool sg[NPAGES * PAGE_DW];
#endif
    pthread_barrier_t bar;
};

struct WaveSim {
    using T = FeP;
    WaveSimShared *sh;
    int p;
    Fe ld1(uint32_t dw) const {
        Fe v;
        for (int i = 0; i < 9; ++i) v.l[i] = sh->limb[dw + 64 * i];
        BN_IFB(v.lb = sh->lb[dw]; v.vb = sh->vb[dw]; v.sg = sh->sg[dw];)
        return v;
    }
    void st1(uint32_t dw, const Fe &v) const {
        for (int i = 0; i < 9; ++i) sh->limb[dw + 64 * i] = v.l[i];
        BN_IFB(sh->lb[dw] = v.lb; sh->vb[dw] = v.vb; sh->sg[dw] = v.sg;)
    }
    FeP ld(uint32_t off) const { return {{ld1(off >> 2), ld1((off >> 2) + 1)}}; }
    void st(uint32_t off, const FeP &v) const { st1(off >> 2, v.v[0]); st1((off >> 2) + 1, v.v[1]); }
    Role role(uint32_t phase) const { return ROLES[phase][p]; }
    int pair() const { return p; }
    void sync() const { pthread_barrier_wait(&sh->bar); }
};

// zero register file (the ZERO register is tracked as lb = 0, vb = 0: a padded gather term adds nothing, also to the bounds),
// Frobenius multipliers in their registers
inline void wavesim_init(WaveSimShared &sh) {
    for (int i = 0; i < NPAGES * PAGE_DW; ++i) {
        sh.limb[i] = 0;
        BN_IFB(sh.lb[i] = 0; sh.vb[i] = 0; sh.sg[i] = false;)
    }
    WaveSim w = {&sh, 0};
    for (int j = 0; j < 18; ++j) {
        FeP c;
        for (int k = 0; k < 2; ++k) { for (int i = 0; i < 9; ++i) c.v[k].l[i] = KCONST[j][k][i]; BN_SETB(c.v[k], 1, 1); }
        w.st((uint32_t)KBASE_OFF[1] + 8u * (uint32_t)j, c);
    }
    for (int j = 0; j < NMCONST; ++j) {
        FeP c;
        for (int k = 0; k < 2; ++k) { for (int i = 0; i < 9; ++i) c.v[k].l[i] = MCONST[j][k][i]; BN_SETB(c.v[k], 1, 1); }
        w.st(MCONST_OFF[j], c);
    }
}

// runs body(w) on 32 threads (one per lane pair)
template <class Fn>
void wavesim_run(WaveSimShared &sh, Fn body) {
    pthread_barrier_init(&sh.bar, nullptr, 32);
    struct Arg { WaveSimShared *sh; int p; Fn *fn; };
    std::vector<Arg> args(32);
    std::vector<pthread_t> th(32);
    for (int p = 0; p < 32; ++p) {
        args[p] = {&sh, p, &body};
        pthread_create(&th[p], nullptr, [](void *a) -> void * { Arg *x = (Arg *)a; WaveSim w = {x->sh, x->p}; (*x->fn)(w); return nullptr; }, &args[p]);
    }
    for (auto &t : th) pthread_join(t, nullptr);
    pthread_barrier_destroy(&sh.bar);
}

}}  // namespace bn254::wv
