// blend_instrument.h -- every measurement / ablation hook of the two blend kernels, in ONE place.
//
// The product build defines none of EGS_MEASURE / EGS_ABL / EGS_NO_LRPT and every hook below compiles to nothing (or to the one
// product expression it stands for): render_fwd.hip and render_bwd.hip contain only hook NAMES, no #if.  The instrumented builds are
// made by tools (tools/lane_use.py, tools/final_profiles.sh:  make OBJDIR=... LIB=... EXTRA=-DEGS_MEASURE=n | -DEGS_ABL=n) and are
// never shipped; ablation builds compute WRONG results on purpose (they time a kernel with a piece removed).
//   EGS_MEASURE  1  forward: quad_work <- (wave, splat) visits          2  forward: quad_work <- kept lanes (sum over visits)
//                3  forward: per-wave timeline into n_contrib           4  backward: per-wave timeline into n_contrib
//   EGS_ABL      1  forward without the visit / pair counters           2  ... and without the last-contributor index
//                3  forward with exp2 replaced by a polynomial stub     4  forward without the LDS prefetch of the next record
//                5  backward without the cross-lane reduction and the atomic (the pair arithmetic alone)
//                6  backward with the atomic replaced by a plain store
//                8  backward publishing nothing       9  backward publishing every second visit (request-count ablations, round 5)
//                7  backward without the accumulation of splats whose box covers >= EGS_ABL7_AREA pixels (hot accumulator lines)
//   EGS_NO_LRPT     no issue-priority steps (s_setprio) in either kernel
#pragma once

#ifndef EGS_ABL
#define EGS_ABL 0
#endif

// ---- both kernels ------------------------------------------------------------------------------------------------------------
#ifdef EGS_NO_LRPT
#define EGS_LRPT(...)
#else
#define EGS_LRPT(...) __VA_ARGS__
#endif
#ifdef EGS_MEASURE
#define EGS_IF_MEASURE(...) __VA_ARGS__
#else
#define EGS_IF_MEASURE(...)
#endif

// ---- forward (render_fwd.hip; the names used are the kernel's locals) -----------------------------------------------------------
#ifdef EGS_MEASURE
#define EGS_FWD_MEAS(W) meas += EGS_MEASURE != 2 ? 1u : (uint32_t)__popcll(__ballot((W) > 0.f));
#else
#define EGS_FWD_MEAS(W)
#endif
#if EGS_ABL == 1 || EGS_ABL == 2
#define EGS_FWD_COUNT_VISIT
#define EGS_FWD_COUNT_PAIRS(USED)
#else
#define EGS_FWD_COUNT_VISIT visits++;
#define EGS_FWD_COUNT_PAIRS(USED) pairs += (uint32_t)__popcll(__ballot(USED));
#endif
#if EGS_ABL == 2
#define EGS_FWD_LAST(USED, J) last = (USED) ? 1u : last;
#else
#define EGS_FWD_LAST(USED, J) last = (USED) ? base + (J) + 1u : last;
#endif
#if EGS_ABL == 3
#define EGS_FWD_ALPHA egs_alpha_noexp
#else
#define EGS_FWD_ALPHA egs_alpha
#endif
#if EGS_ABL == 4
#define EGS_FWD_PREFETCH(FETCH, COPY) COPY
#else
#define EGS_FWD_PREFETCH(FETCH, COPY) FETCH
#endif
#if defined(EGS_MEASURE) && EGS_MEASURE == 3
#define EGS_FWD_TIMELINE()                                                                                             \
    {   uint32_t hw, xcc;                                                                                              \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                               \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                             \
        if (lane == 0) n_contrib[pix] = (uint32_t)t_start;                                                             \
        if (lane == 1) n_contrib[pix] = (uint32_t)wall_clock64();                                                      \
        if (lane == 2) n_contrib[pix] = ((xcc & 0xfu) << 16) | (hw & 0xffffu);                                         \
        if (lane == 3) n_contrib[pix] = n;                                                                             \
        if (lane == 4) n_contrib[pix] = meas;                                                                          \
        uint32_t wm = last;                                                                                            \
        for (int d = 32; d >= 1; d >>= 1) wm = max(wm, (uint32_t)__shfl_xor((int)wm, d, 64));                          \
        if (lane == 5) n_contrib[pix] = wm; }
#else
#define EGS_FWD_TIMELINE()
#endif

// ---- backward (render_bwd.hip) -----------------------------------------------------------------------------------------------------
#if defined(EGS_MEASURE) && EGS_MEASURE == 4
#define EGS_BWD_MEASURE(...) __VA_ARGS__
#define EGS_BWD_TIMELINE()                                                                                             \
    if (inside) {                                                    /* n_contrib was consumed above: reuse it as the log */ \
        uint32_t hw, xcc;                                                                                              \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                               \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                             \
        uint32_t* log = const_cast<uint32_t*>(n_contrib) + (size_t)py * W + px;                                        \
        if (lane == 0) *log = (uint32_t)t_start;                                                                       \
        if (lane == 1) *log = (uint32_t)wall_clock64();                                                                \
        if (lane == 2) *log = ((xcc & 0xfu) << 16) | (hw & 0xffffu);                                                   \
        if (lane == 3) *log = range.y - range.x;                                                                       \
        if (lane == 4) *log = meas;                                                                                    \
        if (lane == 5) *log = wmax; }
#else
#define EGS_BWD_MEASURE(...)
#define EGS_BWD_TIMELINE()
#endif
#if EGS_ABL == 6                       // the reduction kept, the atomic replaced by a plain store to the same word
#define EGS_BWD_ACCUM(PTR, VAL) (*(PTR) = (VAL))
#elif EGS_ABL == 8                     // the reduction kept, NOTHING published (a request-free floor; sums wrong on purpose)
#define EGS_BWD_ACCUM(PTR, VAL) do { if ((VAL) == 12345.678f) *(PTR) = (VAL); } while (0)
#elif EGS_ABL == 9                     // every second visit publishes (half the accumulator requests, all of the arithmetic)
#define EGS_BWD_ACCUM(PTR, VAL) do { if ((j & 1) || (VAL) == 12345.678f) unsafeAtomicAdd((PTR), (VAL)); } while (0)
#else
#define EGS_BWD_ACCUM(PTR, VAL) unsafeAtomicAdd((PTR), (VAL))
#endif
#if EGS_ABL == 7                       // no accumulation at all for splats whose alpha >= 1/255 box covers EGS_ABL7_AREA pixels or more
#define EGS_BWD_ABL7(C2) { const uint32_t bx_ = __float_as_uint((C2).z), by_ = __float_as_uint((C2).w);                          \
                           if ((((bx_ >> 16) & 0x7fffu) - (bx_ & 0x7fffu) + 1u) * (((by_ >> 16) & 0x7fffu) - (by_ & 0x7fffu) + 1u) >= (uint32_t)(EGS_ABL7_AREA)) continue; }
#else
#define EGS_BWD_ABL7(C2)
#endif
#if EGS_ABL == 5
#define EGS_BWD_ABL5(...) __VA_ARGS__
#else
#define EGS_BWD_ABL5(...)
#endif
