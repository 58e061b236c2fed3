// jslp_tu_resident.hip -- the register-resident kernel's INSTANCES as translation units of their own (round 6).
//
// jslp_hip.hip used to be one translation unit with ~90 kernel instantiations, 30 of them k_simplex_resident<...> (7-15 k instructions each):
// 2.5 minutes per library whatever was edited.  The product build compiles jslp_hip.hip with -DJSLP_SPLIT_TU -- host code, the small
// kernels, the node kernels, the fused pipeline; no resident instance -- and this file twice, -DJSLP_TU_PART=1 (the 1024-lane geometries and
// <512,4,8>) and -DJSLP_TU_PART=2 (the tall / wide 512-lane geometries, the XCD-local build of the test library), all three side by side
// (__graft_entry__.build()).  A part exports ONE hidden function that launches the instance a key names, or says "not mine".
// The kernel headers are included inside a namespace of the part's own: every kernel and every helper gets internal linkage, so the parts and
// the main unit never meet at link time; ResCtx travels as bytes (the same struct in every unit: same headers, same flags).
// (A single-unit build -- `hipcc jslp_hip.hip` without -DJSLP_SPLIT_TU, what tools/kernel_resources.py / isa_mix.py and the development
//  builds use -- is still whole: same kernels, same instances.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stddef.h>

#ifndef JSLP_TU_PART
#error "compile with -DJSLP_TU_PART=1 or 2"
#endif
#if defined(JSLP_CHAOS_BUILD) && !defined(JSLP_WITH_XL)
#define JSLP_WITH_XL 1
#endif

namespace {
#include "jslp_kernels.hip.h"

template <int T, int C, int R, bool UNR, bool LEAN, bool OPT, bool CHK, bool XL>
hipError_t launch(unsigned grid, const void* rc_bytes, hipStream_t s) {
    ResCtx rc = *static_cast<const ResCtx*>(rc_bytes);
    void* args[] = {&rc};
    return hipLaunchCooperativeKernel((const void*)k_simplex_resident<T, C, R, UNR, LEAN, OPT, CHK, XL>, dim3(XL ? JSLP_XL_SPREAD * grid : grid), dim3(T), args, 0, s);
}
// lean instances of one geometry: unrestricted variables x cycle check
template <int T, int C, int R>
hipError_t launch_lean(bool unr, bool chk, unsigned grid, const void* rc, hipStream_t s) {
    return unr ? (chk ? launch<T, C, R, true, true, false, true, false>(grid, rc, s) : launch<T, C, R, true, true, false, false, false>(grid, rc, s))
               : (chk ? launch<T, C, R, false, true, false, true, false>(grid, rc, s) : launch<T, C, R, false, true, false, false, false>(grid, rc, s));
}
}  // namespace

// key: {threads, columns per lane, rows per workgroup, unr, lean, opt, chk, xl}.  Returns the hipError_t of the launch, or -1: not an instance of this part.
#if JSLP_TU_PART == 1
extern "C" __attribute__((visibility("hidden"))) int jslpx_resident_launch_1(const int* key, unsigned grid, const void* rc, void* stream) {
#else
extern "C" __attribute__((visibility("hidden"))) int jslpx_resident_launch_2(const int* key, unsigned grid, const void* rc, void* stream) {
#endif
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int T = key[0], C = key[1], R = key[2];
    const bool unr = key[3] != 0, lean = key[4] != 0, opt = key[5] != 0, chk = key[6] != 0, xl = key[7] != 0;
#if JSLP_TU_PART == 1
    if (xl) return -1;
    if (T == 1024 && C == 2 && R == 8) {
        if (opt) return (lean && !unr) ? (int)(chk ? launch<1024, 2, 8, false, true, true, true, false>(grid, rc, s) : launch<1024, 2, 8, false, true, true, false, false>(grid, rc, s)) : -1;
        if (lean) return (int)launch_lean<1024, 2, 8>(unr, chk, grid, rc, s);
        return (int)(unr ? launch<1024, 2, 8, true, false, false, true, false>(grid, rc, s) : launch<1024, 2, 8, false, false, false, true, false>(grid, rc, s));  // (the general build: CHK = true)
    }
    if (T == 512 && C == 4 && R == 8 && !opt) {
        if (lean) return (int)launch_lean<512, 4, 8>(unr, chk, grid, rc, s);
        return (int)(unr ? launch<512, 4, 8, true, false, false, true, false>(grid, rc, s) : launch<512, 4, 8, false, false, false, true, false>(grid, rc, s));
    }
    return -1;
#else
    if (!lean) return -1;  // (the tall / wide geometries have no general build)
#ifdef JSLP_WITH_XL
    if (xl) return (T == 512 && C == 2 && R == 32 && !unr && !opt) ? (int)(chk ? launch<512, 2, 32, false, true, false, true, true>(grid, rc, s) : launch<512, 2, 32, false, true, false, false, true>(grid, rc, s)) : -1;
#else
    if (xl) return -1;
#endif
    if (T == 512 && C == 4 && R == 16) {
        if (opt) return !unr ? (int)(chk ? launch<512, 4, 16, false, true, true, true, false>(grid, rc, s) : launch<512, 4, 16, false, true, true, false, false>(grid, rc, s)) : -1;
        return (int)launch_lean<512, 4, 16>(unr, chk, grid, rc, s);
    }
    if (opt) return -1;
    if (T == 512 && C == 6 && R == 12) return (int)launch_lean<512, 6, 12>(unr, chk, grid, rc, s);
    if (T == 512 && C == 8 && R == 8) return (int)launch_lean<512, 8, 8>(unr, chk, grid, rc, s);
    return -1;
#endif
}
