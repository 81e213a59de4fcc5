// Shared device/host helpers for libmcamd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <mutex>

#include "../../include/mcamd.h"

typedef _Float16 half_t;
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

#define MCAMD_WAVE 64

// Opt a kernel into more than 64 KB of dynamic LDS exactly once per process.  std::call_once: the header promises
// that the entry points are re-entrant across host threads, and a plain `static bool` flag lets a second thread
// launch before the first one's hipFuncSetAttribute has returned.
#define MCAMD_LDS_OPT_IN(kernel, bytes)                                                                       \
    do {                                                                                                      \
        static std::once_flag once_;                                                                          \
        std::call_once(once_, [&] {                                                                           \
            (void)hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                      (int)(bytes));                                                          \
        });                                                                                                   \
    } while (0)

void mcamd_set_error(const char* fmt, ...);

// Tuning switches (MCAMD_* environment variables) are read ONCE per call site, not per launch:
// MCAMD_ENV_INT caches the value until mcamd_reload_config() bumps the generation (tests and A/B runs that change a
// switch inside one process call it; a training step never touches the environment).
extern int g_mcamd_env_generation;
struct McamdEnvSlot {
    int gen = -1, has = 0, val = 0;
    int get(const char* name, int dflt);
};
#define MCAMD_ENV_INT(name, dflt) (([]() -> McamdEnvSlot* { static McamdEnvSlot slot_; return &slot_; }())->get(name, dflt))

#define MCAMD_REQUIRE(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            mcamd_set_error(__VA_ARGS__);   \
            return MCAMD_EINVAL;            \
        }                                   \
    } while (0)

#define MCAMD_LAUNCH_CHECK(what)                                                    \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            mcamd_set_error("%s: launch failed: %s", what, hipGetErrorString(e_));  \
            return MCAMD_ELAUNCH;                                                   \
        }                                                                           \
    } while (0)

// Launch plans (plan.hip): while a host thread records, every launch-type entry point appends a by-value copy of its
// arguments to the plan instead of launching.
#include <functional>
struct mcamd_plan;
extern thread_local mcamd_plan* g_mcamd_rec;
static inline bool mcamd_recording() { return g_mcamd_rec != nullptr; }
int mcamd_rec_push(void* stream, std::function<int(void*)> fn);

static inline int round_up_int(int v, int m) { return (v + m - 1) / m * m; }
static inline long long round_up_ll(long long v, long long m) { return (v + m - 1) / m * m; }

// Row r of a 32x32 MFMA accumulator fragment held by `lane` in register `reg`
// (cdna_hip_programming.md section 3: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)).
__device__ __forceinline__ int mfma32_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// 16-byte async copy global -> LDS (global_load_lds_dwordx4).  The LDS address must be
// wave-uniform; lane i lands at lds + 16*i.  The global address is per lane.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// XOR swizzle of the 16-byte chunks of an LDS row of CPR chunks (ds_read_b128 operand tiles).
template <int CPR>
__device__ __forceinline__ int swz(int row) {
    // CPR = 16-byte chunks per LDS row.  64-byte rows: 4 rows span the 64 banks -> XOR with
    // (row>>2)&3; 128-byte rows: 2 rows span them -> XOR with (row>>1)&7.  Either makes the 16
    // lanes of every ds_read_b128 lane group hit 16 distinct 16-byte slots.
    return CPR == 4 ? ((row >> 2) & 3) : ((row >> 1) & 7);
}

// Scales of the fp8 correction operands (mcamd_conv_geom.x_f8, mcamd_act_desc.planes == 4, mcamd_pack_job.split == 2):
//   lo8 = e4m3(x_lo * 2^F8_SXL),  x8 = e4m3(x * 2^F8_SX8),  w8 = e4m3(w_hi * 2^wexp),  wlo8 = e4m3(w_lo * 2^(wexp + 11))
// lo8 * w8 and x8 * wlo8 then carry the same factor 2^(F8_SXL + wexp), which the block-scaled MFMA takes back through
// its e8m0 scale operands.  `wexp` is a per-LAYER exponent (mcamd_pack_job.f8_wexp = mcamd_conv_geom.x_f8_wexp; the engine
// derives it from the layer's largest weight so that it lands at 112-224 of e4m3's 448: BatchNorm makes the scale of a
// layer's weights arbitrary, a static exponent served a 30x range either way).  Activations: |x| <= 224 before
// the e4m3 maximum clamps (the correction of such an entry is then partly lost: never worse than plain fp16 operands);
// three mantissa bits down to |x| 2^-7 (x8, lo8).
#define MCAMD_F8_SXL 12
#define MCAMD_F8_SX8 1
#define MCAMD_F8_WEXP_DEFAULT 5      /* |w| <= 14: initialisation-sized weights */
static_assert(MCAMD_F8_SXL == MCAMD_F8_SX8 + 11, "one scale for both correction terms: w_lo is scaled 2^11 above w_hi");
