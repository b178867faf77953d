// alac_unpc.hpp — lane-serial unpc_block forms (codec/dp_dec.c:55-381), in place over a row with element
// stride `rs`; shared by both decode pipelines (alac_decode.hip, alac_decode_v1.hip).
#pragma once

#include "alac_dev.hpp"

namespace alacdev {

// ---- unpc_block, in place over a strided row --------------------------------------------------

template <int NA>
__device__ __forceinline__ void unpc_fixed(int32_t *row, uint64_t rs, uint32_t num, const int16_t *coefs,
                                           uint32_t chanshift, uint32_t denshift)
{
    Lms<NA> s;
#pragma unroll
    for (int k = 0; k < NA; k++) s.a[k] = coefs[k];
#pragma unroll
    for (int k = 0; k <= NA; k++) s.h[k] = 0;
    int32_t dn = num ? row[0] : 0;
    for (uint32_t j = 0; j < num; j++) {
        const int32_t del = dn;
        if (j + 1 < num) dn = row[(uint64_t)(j + 1) * rs];
        int32_t out;
        if (j == 0) {
            out = del;
            lms_push<NA>(s, out);
        } else if (j <= (uint32_t)NA) {
            out = sext(wadd(del, s.h[0]), chanshift);
            lms_push<NA>(s, out);
        } else {
            out = lms_step_dec<NA>(s, del, chanshift, denshift);
        }
        row[(uint64_t)j * rs] = out;
    }
}

// any tap count 1..30 (codec/dp_dec.c:335-380); history and coefficients in private memory
__device__ __noinline__ inline void unpc_general(int32_t *row, uint64_t rs, uint32_t num, const int16_t *coefs, uint32_t na,
                             uint32_t chanshift, uint32_t denshift, int16_t *coefsOut = nullptr)
{
    int32_t a[32], h[33];
    for (uint32_t k = 0; k < 32; k++) a[k] = k < na ? coefs[k] : 0;
    for (uint32_t k = 0; k < 33; k++) h[k] = 0;
    const int32_t denhalf = denshift ? (1 << (denshift - 1)) : 0;
    for (uint32_t j = 0; j < num; j++) {
        const int32_t del = row[(uint64_t)j * rs];
        int32_t out;
        if (j == 0) {
            out = del;
        } else if (j <= na) {
            out = sext(wadd(del, h[0]), chanshift);
        } else {
            const int32_t top = h[na];
            int32_t sum = 0;
            for (uint32_t k = 0; k < na; k++) sum = wadd(sum, wmul(a[k], wsub(h[k], top)));
            out = sext(wadd(wadd(del, top), wadd(sum, denhalf) >> denshift), chanshift);
            const int32_t sg = sign_of(del);
            int32_t del0 = del;
            if (sg > 0) {
                for (int32_t k = (int32_t)na - 1; k >= 0; k--) {
                    const int32_t dd = wsub(top, h[k]);
                    const int32_t sgn = sign_of(dd);
                    a[k] = (int16_t)(a[k] - sgn);
                    del0 = wsub(del0, wmul((int32_t)na - k, wmul(sgn, dd) >> denshift));
                    if (del0 <= 0) break;
                }
            } else if (sg < 0) {
                for (int32_t k = (int32_t)na - 1; k >= 0; k--) {
                    const int32_t dd = wsub(top, h[k]);
                    const int32_t sgn = sign_of(dd);
                    a[k] = (int16_t)(a[k] + sgn);
                    del0 = wsub(del0, wmul((int32_t)na - k, wmul(-sgn, dd) >> denshift));
                    if (del0 >= 0) break;
                }
            }
        }
        for (uint32_t k = na; k > 0; k--) h[k] = h[k - 1];
        h[0] = out;
        row[(uint64_t)j * rs] = out;
    }
    if (coefsOut)
        for (uint32_t k = 0; k < na; k++) coefsOut[k] = (int16_t)a[k];
}

// numactive == 31: first-order (codec/dp_dec.c:74-95)
__device__ __forceinline__ void unpc_first_order(int32_t *row, uint64_t rs, uint32_t num, uint32_t chanshift)
{
    if (!num) return;
    int32_t prev = row[0];
    for (uint32_t j = 1; j < num; j++) {
        prev = sext(wadd(row[(uint64_t)j * rs], prev), chanshift);
        row[(uint64_t)j * rs] = prev;
    }
}

__device__ __forceinline__ void unpc_any(int32_t *row, uint64_t rs, uint32_t num, const int16_t *coefs,
                                         uint32_t na, uint32_t chanbits, uint32_t denshift)
{
    const uint32_t chanshift = 32 - chanbits;
    if (na == 0) return;  // copy, in place
    if (na == 31)
        unpc_first_order(row, rs, num, chanshift);
    else if (na == 4)
        unpc_fixed<4>(row, rs, num, coefs, chanshift, denshift);
    else if (na == 8)
        unpc_fixed<8>(row, rs, num, coefs, chanshift, denshift);
    else
        unpc_general(row, rs, num, coefs, na, chanshift, denshift);
}


}  // namespace alacdev
