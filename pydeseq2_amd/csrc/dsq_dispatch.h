// dsq_dispatch.h — compile-time design width P (number of design-matrix columns).
// Per-gene routines keep the p x p normal equations in registers, so P must be a
// template parameter; the runtime value is dispatched here.
#pragma once

#define DSQ_REG_MAX_P 12  // widest design of the register / cell kernels; wider ones: dsq_wide.h

// developer builds (tools/asm_blocks.py): -DDSQ_ONLY_P=8 instantiates a single width, a unit compiles in seconds
#if defined(DSQ_ONLY_P)
#define DSQ_P_CASE(N_, ...)                 \
    case N_: {                              \
        constexpr int P = N_;               \
        if constexpr (N_ == DSQ_ONLY_P) {   \
            __VA_ARGS__;                    \
        }                                   \
    } break;
#else
#define DSQ_P_CASE(N_, ...) \
    case N_: {              \
        constexpr int P = N_; \
        __VA_ARGS__;        \
    } break;
#endif

#define DSQ_DISPATCH_P(p_, ...)                  \
    switch (p_) {                                \
        DSQ_P_CASE(1, __VA_ARGS__)               \
        DSQ_P_CASE(2, __VA_ARGS__)               \
        DSQ_P_CASE(3, __VA_ARGS__)               \
        DSQ_P_CASE(4, __VA_ARGS__)               \
        DSQ_P_CASE(5, __VA_ARGS__)               \
        DSQ_P_CASE(6, __VA_ARGS__)               \
        DSQ_P_CASE(7, __VA_ARGS__)               \
        DSQ_P_CASE(8, __VA_ARGS__)               \
        DSQ_P_CASE(9, __VA_ARGS__)               \
        DSQ_P_CASE(10, __VA_ARGS__)              \
        DSQ_P_CASE(11, __VA_ARGS__)              \
        DSQ_P_CASE(12, __VA_ARGS__)              \
        default: break;                          \
    }
