// alac_decode.hip — batch ALAC decode kernels + the batched stage-level entry points.
//
// Decode pipeline (packets are independent: coefficients travel in each header):
//   k_decode_entropy  one lane per packet: element/header parse + dyn_decomp of U then V (V starts
//                     where U's bits end, so the two channels of a packet are serial here)
//   k_decode_unpc     one lane per (packet, channel): unpc_block in place
//   k_decode_unmix    LDS-transposed un-mix + PCM packing, coalesced on both sides
//
// Reference: ALACDecoder::Decode codec/ALACDecoder.cu:571-1002, fillWriteBuffer :497-563,
// dyn_decomp codec/ag_dec.c:272-362, unpc_block codec/dp_dec.c:55-381.
#include <algorithm>
#include <cstdlib>
#include "alac_dev.hpp"
#include "alac_kernels.hpp"
#include "alac_unpc.hpp"

namespace alacdev {

// ------------------------------------------------------------------------------------------------
// dyn_decomp (codec/ag_dec.c:272-362; dyn_get_32bit :220-270, dyn_get :171-217), lane-serial.
// Writes residual c to dst[c * dstStride].  Returns status; *bitsUsed = consumed bits.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t decomp_lane(const uint8_t *base, uint64_t nbytes, uint64_t &pos,
                                               int32_t *dst, uint64_t dstStride, uint32_t numSamples,
                                               uint32_t maxSize, uint32_t mb0, uint32_t pb, uint32_t kb)
{
    const uint32_t wb = (1u << kb) - 1;
    const uint64_t maxPos = nbytes * 8;
    uint32_t mb = mb0, zmode = 0, c = 0;
    int32_t status = 0;
    while (c < numSamples) {
        if (!(pos < maxPos)) {  // :302
            status = -50;
            break;
        }
        uint32_t m = mb >> kQBShift;
        uint32_t k = (uint32_t)lg3a(m);
        k = k > kb ? kb : k;
        m = (1u << k) - 1;

        uint32_t n;
        {
            uint32_t streamlong = peek32(base, nbytes, pos);
            const uint32_t pre = (uint32_t)lead(~streamlong);
            if (pre >= kMaxPrefix) {
                uint64_t q = pos + kMaxPrefix;
                n = read_bits(base, nbytes, q, maxSize);
                pos += kMaxPrefix + maxSize;
            } else {
                pos += pre + 1;
                n = pre;
                if (k != 1) {
                    streamlong <<= pre + 1;
                    const uint32_t v = streamlong >> (32 - k);
                    pos += k - 1;
                    n = pre * m;
                    if (v >= 2) {
                        n += v - 1;
                        pos += 1;
                    }
                }
            }
        }
        const uint32_t nd = n + zmode;
        const int32_t mult = (-(int32_t)(nd & 1)) | 1;
        dst[(uint64_t)c * dstStride] = (int32_t)((nd + 1) >> 1) * mult;
        c++;

        mb = pb * (n + zmode) + mb - ((pb * mb) >> kQBShift);
        if (n > kMeanClamp) mb = kMeanClamp;
        zmode = 0;

        if (((mb << 2) < (1u << kQBShift)) && (c < numSamples)) {
            zmode = 1;
            k = (uint32_t)(lead(mb) - 24 + (int32_t)((mb + 16u) >> 6));
            const uint32_t mz = ((1u << k) - 1) & wb;
            uint32_t streamlong = peek32(base, nbytes, pos);
            const uint32_t pre = (uint32_t)lead(~streamlong);
            uint32_t nz;
            if (pre >= kMaxPrefix) {
                streamlong <<= kMaxPrefix;
                nz = streamlong >> (32 - kMaxRunBits);
                pos += kMaxPrefix + kMaxRunBits;
            } else {
                pos += pre + 1;
                streamlong <<= pre + 1;
                const uint32_t v = streamlong >> (32 - k);
                pos += k;
                nz = pre * mz + v - 1;
                if (v < 2) {
                    nz -= (v - 1);
                    pos -= 1;
                }
            }
            if (!((uint64_t)c + nz <= (uint64_t)numSamples)) {  // :341
                status = -50;
                break;
            }
            for (uint32_t j = 0; j < nz; j++) {
                dst[(uint64_t)c * dstStride] = 0;
                c++;
            }
            if (nz >= 65535) zmode = 0;
            mb = 0;
        }
    }
    if (status == 0 && (pos + 7) / 8 > nbytes) status = -50;  // :359
    return status;
}

__global__ __launch_bounds__(64) void k_decode_entropy(DecodeArgs A)
{
    if (A.gate && *A.gate == 0) return;
    const uint32_t p = blockIdx.x * 64u + threadIdx.x;
    if (p >= A.numPackets) return;
    const uint64_t off = A.offsets[p];
    const uint64_t nbytes = A.offsets[p + 1] - off;
    const uint8_t *base = A.stream + off;
    const uint64_t rs = A.numPackets;  // residual stride between consecutive samples
    // one record per element (codec/ALACDecoder.cu:600-990 loops over the elements of a packet until every channel
    // of the cookie is decoded or ID_END is met); record 0 also carries the packet's status
    uint32_t elem = 0, channelIndex = 0;
    for (uint32_t e = 0; e < A.maxElems; e++) {
        DecRec *r = A.recs + (uint64_t)e * A.numPackets + p;
        r->numSamples = 0;
        r->escape = 0;
        r->mixBits = r->mixRes = 0;
        r->bytesShifted = 0;
        r->elementChannels = 0;
        r->shiftPos = 0;
        r->chanIndex = 0;
    }
    DecRec *rec = A.recs + p;
    uint32_t lastSamples = 0;

    uint64_t pos = 0;
    uint32_t numSamples = A.frameSize;
    int32_t status = 0;
    bool done = false;

    while (!done && status == 0) {
        if (!((pos >> 3) < nbytes)) {  // :615
            status = -50;
            break;
        }
        const uint32_t tag = read_bits(base, nbytes, pos, 3);
        switch (tag) {
        case 0:    // ID_SCE
        case 3:    // ID_LFE
        case 1: {  // ID_CPE
            const uint32_t ech = (tag == 1) ? 2u : 1u;
            if (channelIndex + ech > A.numChannels || elem >= A.maxElems) {  // :760-762 (a pair that does not fit ends the packet)
                done = true;
                break;
            }
            rec = A.recs + (uint64_t)elem * A.numPackets + p;
            int32_t *resU = A.resid + (uint64_t)channelIndex * A.frameSize * rs + p;
            int32_t *resV = resU + (uint64_t)A.frameSize * rs;
            (void)read_bits(base, nbytes, pos, 4);
            if (read_bits(base, nbytes, pos, 12) != 0) {  // :633 / :768
                status = -50;
                break;
            }
            const uint32_t hb = read_bits(base, nbytes, pos, 4);
            const uint32_t partial = hb >> 3, shb = (hb >> 1) & 3, esc = hb & 1;
            if (shb == 3) {
                status = -50;
                break;
            }
            uint32_t chanBits = A.bitDepth - shb * 8 + (ech == 2 ? 1 : 0);
            if (partial) numSamples = read_bits(base, nbytes, pos, 32);
            if (numSamples > A.frameSize) {
                status = -50;
                break;
            }
            rec->elementChannels = ech;
            rec->chanIndex = channelIndex;
            if (!esc) {
                const uint32_t mixBits = read_bits(base, nbytes, pos, 8);
                const int32_t mixRes = (int8_t)read_bits(base, nbytes, pos, 8);
                for (uint32_t c = 0; c < ech; c++) {
                    uint32_t b = read_bits(base, nbytes, pos, 8);
                    rec->c[c].mode = (uint16_t)(b >> 4);
                    rec->c[c].denShift = (uint16_t)(b & 0xf);
                    b = read_bits(base, nbytes, pos, 8);
                    rec->c[c].pbFactor = (uint16_t)(b >> 5);
                    rec->c[c].num = (uint16_t)(b & 0x1f);
                    for (uint32_t i = 0; i < (b & 0x1f); i++)
                        rec->c[c].coefs[i] = (int16_t)read_bits(base, nbytes, pos, 16);
                }
                rec->shiftPos = pos;
                if (shb) pos += (uint64_t)shb * 8 * ech * numSamples;
                status = decomp_lane(base, nbytes, pos, resU, rs, numSamples, chanBits, A.mb,
                                     (A.pb * rec->c[0].pbFactor) / 4, A.kb);
                if (status == 0 && ech == 2)
                    status = decomp_lane(base, nbytes, pos, resV, rs, numSamples, chanBits, A.mb,
                                         (A.pb * rec->c[1].pbFactor) / 4, A.kb);
                rec->mixBits = (int32_t)mixBits;
                rec->mixRes = mixRes;
                rec->bytesShifted = shb;
            } else {
                // :697-727 / :856-896 uncompressed element
                chanBits = A.bitDepth;
                const uint32_t sh = 32 - chanBits;
                for (uint32_t i = 0; i < numSamples; i++) {
                    resU[(uint64_t)i * rs] = (int32_t)(read_bits(base, nbytes, pos, chanBits) << sh) >> sh;
                    if (ech == 2)
                        resV[(uint64_t)i * rs] = (int32_t)(read_bits(base, nbytes, pos, chanBits) << sh) >> sh;
                }
                rec->escape = 1;
                if ((pos + 7) / 8 > nbytes) status = -50;  // the fixed-width payload ran past the packet (:996-1000)
            }
            rec->numSamples = numSamples;
            lastSamples = numSamples;
            channelIndex += ech;
            elem++;
            done = channelIndex >= A.numChannels;  // :967
            break;
        }
        case 2:  // ID_CCE
        case 5:  // ID_PCE
            status = -50;
            break;
        case 4: {  // ID_DSE :1033-1059
            (void)read_bits(base, nbytes, pos, 4);
            const uint32_t align = read_bits(base, nbytes, pos, 1);
            uint32_t count = read_bits(base, nbytes, pos, 8);
            if (count == 255) count += read_bits(base, nbytes, pos, 8);
            if (align && (pos & 7)) pos += 8 - (pos & 7);
            pos += (uint64_t)count * 8;
            if ((pos + 7) / 8 > nbytes) status = -50;
            break;
        }
        case 6: {  // ID_FIL :1012-1027
            int32_t count = (int32_t)read_bits(base, nbytes, pos, 4);
            if (count == 15) count += (int32_t)read_bits(base, nbytes, pos, 8) - 1;
            pos += (uint64_t)count * 8;
            if ((pos + 7) / 8 > nbytes) status = -50;
            break;
        }
        default:  // ID_END before any audio element
            done = true;
            break;
        }
    }
    A.recs[p].status = status;
    A.statusOut[p] = status;
    A.numSamplesOut[p] = status == 0 ? lastSamples : 0;
}

// the element record that carries output channel ch of packet p (nullptr: the packet has none for it)
__device__ __forceinline__ const DecRec *element_of(const DecodeArgs &A, uint32_t p, uint32_t ch)
{
    for (uint32_t e = 0; e < A.maxElems; e++) {
        const DecRec *r = A.recs + (uint64_t)e * A.numPackets + p;
        if (r->elementChannels == 0) break;
        if (ch >= r->chanIndex && ch < r->chanIndex + r->elementChannels) return r;
    }
    return nullptr;
}

__global__ __launch_bounds__(64) void k_decode_unpc(DecodeArgs A)
{
    if (A.gate && *A.gate == 0) return;
    const uint64_t gid = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    if (gid >= (uint64_t)A.numPackets * A.numChannels) return;
    const uint32_t ch = (uint32_t)(gid / A.numPackets);
    const uint32_t p = (uint32_t)(gid % A.numPackets);
    if (A.recs[p].status != 0) return;
    const DecRec *rec = element_of(A, p, ch);
    if (!rec || rec->escape) return;
    const uint32_t chanbits = A.bitDepth - rec->bytesShifted * 8 + (rec->elementChannels == 2 ? 1 : 0);
    int32_t *row = A.resid + (uint64_t)ch * A.frameSize * A.numPackets + p;
    const DecChan &c = rec->c[ch - rec->chanIndex];
    // :829-838: mode != 0 runs the first-order pass first
    if (c.mode != 0) unpc_first_order(row, A.numPackets, rec->numSamples, 32 - chanbits);
    unpc_any(row, A.numPackets, rec->numSamples, c.coefs, c.num, chanbits, c.denShift);
}

// ---- un-mix + pack (gpu_unmixNN / gpu_copyPredictorToNN, codec/ALACDecoder.cu:193-495) --------

template <int DEPTH>
__device__ __forceinline__ void store_sample(uint8_t *p, int32_t x)
{
    if constexpr (DEPTH == 16) {
        *(int16_t *)p = (int16_t)x;
    } else if constexpr (DEPTH == 32) {
        *(int32_t *)p = x;
    } else {
        if constexpr (DEPTH == 20) x = (int32_t)((uint32_t)x << 4);
        p[0] = (uint8_t)x;
        p[1] = (uint8_t)(x >> 8);
        p[2] = (uint8_t)(x >> 16);
    }
}

template <int DEPTH, int CH>
__global__ __launch_bounds__(256) void k_decode_unmix(DecodeArgs A)
{
    if (A.gate && *A.gate == 0) return;
    __shared__ int32_t tu[64][65];
    __shared__ int32_t tv[CH == 2 ? 64 : 1][65];
    const uint32_t lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    // tile = 64 packets x 64 samples; the plain launch has one workgroup per tile (2-D grid), the gated one (the lane decoder
    // behind the fast pipeline, nothing to do unless some packet carried another element sequence) a small 1-D grid that walks
    // the tiles: 500 000 workgroups that read the gate and left cost 28 us of every 125 000-packet decode
    const uint32_t nPx = (A.numPackets + 63) / 64;
    const uint64_t tiles = (uint64_t)nPx * ((A.frameSize + 63) / 64);
    for (uint64_t tile = blockIdx.x + (uint64_t)gridDim.x * blockIdx.y; tile < tiles; tile += (uint64_t)gridDim.x * gridDim.y) {
    const uint32_t tileP = (uint32_t)(tile % nPx) * 64u, tileJ = (uint32_t)(tile / nPx) * 64u;

    // load: consecutive lanes = consecutive packets (the residual layout's fast axis)
    for (uint32_t i = 0; i < 16; i++) {
        const uint32_t jj = grp + 4 * i;
        const uint32_t j = tileJ + jj, p = tileP + lane;
        int32_t u = 0, v = 0;
        if (p < A.numPackets && j < A.frameSize) {
            u = A.resid[(uint64_t)j * A.numPackets + p];
            if constexpr (CH == 2) v = A.resid[((uint64_t)A.frameSize + j) * A.numPackets + p];
        }
        tu[jj][lane] = u;
        if constexpr (CH == 2) tv[jj][lane] = v;
    }
    __syncthreads();

    constexpr uint32_t BPS = bytes_per_sample(DEPTH);
    for (uint32_t i = 0; i < 16; i++) {
        const uint32_t pp = grp + 4 * i;
        const uint32_t p = tileP + pp, j = tileJ + lane;
        if (p >= A.numPackets) continue;
        const DecRec *rec = A.recs + p;
        if (rec->status != 0 || j >= rec->numSamples) continue;
        const uint32_t shb = rec->bytesShifted;
        uint8_t *op = A.pcmOut + ((uint64_t)p * A.frameSize + j) * CH * BPS;
        int32_t l, r = 0;
        if constexpr (CH == 2) {
            const int32_t u = tu[lane][pp], v = tv[lane][pp];
            if (rec->mixRes != 0) {
                l = u + v - ((rec->mixRes * v) >> rec->mixBits);
                r = l - v;
            } else {
                l = u;
                r = v;
            }
        } else {
            l = tu[lane][pp];
        }
        if (shb != 0 && DEPTH >= 24) {
            const uint8_t *base = A.stream + A.offsets[p];
            const uint64_t nbytes = A.offsets[p + 1] - A.offsets[p];
            uint64_t sp = rec->shiftPos + (uint64_t)j * CH * shb * 8;
            l = (int32_t)(((uint32_t)l << (shb * 8)) | read_bits(base, nbytes, sp, shb * 8));
            if constexpr (CH == 2) r = (int32_t)(((uint32_t)r << (shb * 8)) | read_bits(base, nbytes, sp, shb * 8));
        }
        store_sample<DEPTH>(op, l);
        if constexpr (CH == 2) store_sample<DEPTH>(op + BPS, r);
    }
    __syncthreads();
    }
}

// > 2 channels: the same tile walk once per output channel c; a packet's element that STARTS at c is un-mixed and
// written at channel c (and c + 1) of the numChannels-interleaved frame (unmixNN / copyPredictorToNN with stride
// numChannels, codec/ALACDecoder.cu:733-753,:900-935); channels no element carries are zero (:971-998).
template <int DEPTH>
__global__ __launch_bounds__(256) void k_decode_unmix_mc(DecodeArgs A)
{
    if (A.gate && *A.gate == 0) return;
    __shared__ int32_t tu[64][65];
    __shared__ int32_t tv[64][65];
    const uint32_t lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const uint32_t nPx = (A.numPackets + 63) / 64;
    const uint64_t tiles = (uint64_t)nPx * ((A.frameSize + 63) / 64);
    for (uint64_t tile = blockIdx.x + (uint64_t)gridDim.x * blockIdx.y; tile < tiles; tile += (uint64_t)gridDim.x * gridDim.y) {
    const uint32_t tileP = (uint32_t)(tile % nPx) * 64u, tileJ = (uint32_t)(tile / nPx) * 64u;
    constexpr uint32_t BPS = bytes_per_sample(DEPTH);
    const uint32_t nch = A.numChannels;
    for (uint32_t c = 0; c < nch; c++) {
        for (uint32_t i = 0; i < 16; i++) {
            const uint32_t jj = grp + 4 * i;
            const uint32_t j = tileJ + jj, p = tileP + lane;
            int32_t u = 0, v = 0;
            if (p < A.numPackets && j < A.frameSize) {
                u = A.resid[((uint64_t)c * A.frameSize + j) * A.numPackets + p];
                if (c + 1 < nch) v = A.resid[((uint64_t)(c + 1) * A.frameSize + j) * A.numPackets + p];
            }
            tu[jj][lane] = u;
            tv[jj][lane] = v;
        }
        __syncthreads();
        for (uint32_t i = 0; i < 16; i++) {
            const uint32_t pp = grp + 4 * i;
            const uint32_t p = tileP + pp, j = tileJ + lane;
            if (p >= A.numPackets || A.recs[p].status != 0) continue;
            const DecRec *rec = element_of(A, p, c);
            uint8_t *op = A.pcmOut + (((uint64_t)p * A.frameSize + j) * nch + c) * BPS;
            if (!rec) {
                const uint32_t ns = A.recs[p].elementChannels ? A.recs[p].numSamples : A.frameSize;
                if (j < ns) store_sample<DEPTH>(op, 0);
                continue;
            }
            if (rec->chanIndex != c || j >= rec->numSamples) continue;
            const uint32_t shb = rec->bytesShifted, ech = rec->elementChannels;
            int32_t l, r = 0;
            const int32_t u = tu[lane][pp], v = tv[lane][pp];
            if (ech == 2 && rec->mixRes != 0) {
                l = u + v - ((rec->mixRes * v) >> rec->mixBits);
                r = l - v;
            } else {
                l = u;
                r = v;
            }
            if (shb != 0 && DEPTH >= 24) {
                const uint8_t *base = A.stream + A.offsets[p];
                const uint64_t nbytes = A.offsets[p + 1] - A.offsets[p];
                uint64_t sp = rec->shiftPos + (uint64_t)j * ech * shb * 8;
                l = (int32_t)(((uint32_t)l << (shb * 8)) | read_bits(base, nbytes, sp, shb * 8));
                if (ech == 2) r = (int32_t)(((uint32_t)r << (shb * 8)) | read_bits(base, nbytes, sp, shb * 8));
            }
            store_sample<DEPTH>(op, l);
            if (ech == 2) store_sample<DEPTH>(op + BPS, r);
        }
        __syncthreads();
    }
    }
}

template <int DEPTH>
static void launch_unmix_depth(const DecodeArgs &da, hipStream_t st)
{
    dim3 grid((da.numPackets + 63) / 64, (da.frameSize + 63) / 64);
    if (da.gate) grid = dim3((uint32_t)std::min<uint64_t>((uint64_t)grid.x * grid.y, 2048), 1);  // (see k_decode_unmix)
    // two channels may arrive as one CPE or as two SCE / LFE elements (codec/ALACDecoder.cu:622-756): the per-element
    // kernel follows the records, k_decode_unmix<., 2> would take the packet for one pair
    if (da.numChannels >= 2)
        hipLaunchKernelGGL((k_decode_unmix_mc<DEPTH>), grid, dim3(256), 0, st, da);
    else
        hipLaunchKernelGGL((k_decode_unmix<DEPTH, 1>), grid, dim3(256), 0, st, da);
}

hipError_t launch_decode(const DecodeArgs &da, hipStream_t st)
{
    if (da.numPackets == 0) return hipSuccess;
    hipLaunchKernelGGL(k_decode_entropy, dim3((da.numPackets + 63) / 64), dim3(64), 0, st, da);
    const uint64_t lanes = (uint64_t)da.numPackets * da.numChannels;
    hipLaunchKernelGGL(k_decode_unpc, dim3((uint32_t)((lanes + 63) / 64)), dim3(64), 0, st, da);
    switch (da.bitDepth) {
    case 16: launch_unmix_depth<16>(da, st); break;
    case 20: launch_unmix_depth<20>(da, st); break;
    case 24: launch_unmix_depth<24>(da, st); break;
    case 32: launch_unmix_depth<32>(da, st); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// stage-level batched entry points (one lane per row)
// ------------------------------------------------------------------------------------------------

// pc_block for any numactive (codec/dp_enc.c:77-388), general loop
__device__ void pc_general(const int32_t *in, int32_t *pc, int32_t num, int16_t *coefs, int32_t na,
                           uint32_t chanbits, uint32_t denshift)
{
    const uint32_t chanshift = 32 - chanbits;
    const int32_t denhalf = denshift ? (1 << (denshift - 1)) : 0;
    pc[0] = in[0];
    if (na == 0) {
        for (int32_t j = 1; j < num; j++) pc[j] = in[j];
        return;
    }
    if (na == 31) {
        for (int32_t j = 1; j < num; j++) pc[j] = sext(wsub(in[j], in[j - 1]), chanshift);
        return;
    }
    for (int32_t j = 1; j <= na; j++) pc[j] = sext(wsub(in[j], in[j - 1]), chanshift);
    int32_t a[32];
    for (int32_t k = 0; k < 32; k++) a[k] = k < na ? coefs[k] : 0;
    for (int32_t j = na + 1; j < num; j++) {
        const int32_t top = in[j - na - 1];
        const int32_t *pin = in + j - 1;
        int32_t sum = 0;
        for (int32_t k = 0; k < na; k++) sum = wsub(sum, wmul(a[k], wsub(top, pin[-k])));
        const int32_t del = sext(wsub(wsub(in[j], top), wadd(sum, denhalf) >> denshift), chanshift);
        pc[j] = del;
        int32_t del0 = del;
        const int32_t sg = sign_of(del);
        if (sg > 0) {
            for (int32_t k = na - 1; k >= 0; k--) {
                const int32_t dd = wsub(top, pin[-k]);
                const int32_t sgn = sign_of(dd);
                a[k] = (int16_t)(a[k] - sgn);
                del0 = wsub(del0, wmul(na - k, wmul(sgn, dd) >> denshift));
                if (del0 <= 0) break;
            }
        } else if (sg < 0) {
            for (int32_t k = na - 1; k >= 0; k--) {
                const int32_t dd = wsub(top, pin[-k]);
                const int32_t sgn = sign_of(dd);
                a[k] = (int16_t)(a[k] + sgn);
                del0 = wsub(del0, wmul(na - k, wmul(-sgn, dd) >> denshift));
                if (del0 >= 0) break;
            }
        }
    }
    for (int32_t k = 0; k < na; k++) coefs[k] = (int16_t)a[k];
}

__global__ __launch_bounds__(64) void k_pc_block(const int32_t *in, int32_t *pc, uint32_t rows, uint32_t stride,
                                                 int32_t num, int16_t *coefs, int32_t na, uint32_t chanbits,
                                                 uint32_t denshift)
{
    const uint32_t r = blockIdx.x * 64u + threadIdx.x;
    if (r >= rows) return;
    pc_general(in + (uint64_t)r * stride, pc + (uint64_t)r * stride, num, coefs + (uint64_t)r * 32, na, chanbits,
               denshift);
}

__global__ __launch_bounds__(64) void k_unpc_block(const int32_t *pc, int32_t *out, uint32_t rows, uint32_t stride,
                                                   int32_t num, int16_t *coefs, int32_t na, uint32_t chanbits,
                                                   uint32_t denshift)
{
    const uint32_t r = blockIdx.x * 64u + threadIdx.x;
    if (r >= rows) return;
    const int32_t *src = pc + (uint64_t)r * stride;
    int32_t *dst = out + (uint64_t)r * stride;
    if (src != dst)
        for (int32_t j = 0; j < num; j++) dst[j] = src[j];
    int16_t *c = coefs + (uint64_t)r * 32;
    if (na == 31)
        unpc_first_order(dst, 1, (uint32_t)num, 32 - chanbits);
    else if (na != 0)
        unpc_general(dst, 1, (uint32_t)num, c, (uint32_t)na, 32 - chanbits, denshift, c);
}

__global__ __launch_bounds__(64) void k_dyn_comp(uint32_t mb0, uint32_t pb, uint32_t kb, const int32_t *pc,
                                                 uint32_t rows, uint32_t stride, int32_t numSamples, int32_t bitSize,
                                                 uint32_t *words, uint32_t wcap, uint32_t *numBits)
{
    const uint32_t r = blockIdx.x * 64u + threadIdx.x;
    if (r >= rows) return;
    const int32_t *src = pc + (uint64_t)r * stride;
    Golomb g;
    gol_reset(g, mb0, pb, kb);
    g.wp = words ? words + (uint64_t)r * wcap : nullptr;
    g.wcap = words ? wcap : 0;
    for (int32_t j = 0; j < numSamples; j++) gol_sym<true>(g, src[j], j + 1 == numSamples, (uint32_t)bitSize);
    gol_flush<true>(g);
    numBits[r] = g.bits;
}

// words (MSB-first uint32) -> bytes in place: byte-swap each word
__global__ void k_bswap_words(uint32_t *w, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = __builtin_bswap32(w[i]);
}

__global__ __launch_bounds__(64) void k_dyn_decomp(uint32_t mb0, uint32_t pb, uint32_t kb, const uint8_t *bits,
                                                   uint32_t bytesStride, uint32_t rows, int32_t *pc, uint32_t stride,
                                                   int32_t numSamples, int32_t maxSize, uint32_t *numBits,
                                                   int32_t *status)
{
    const uint32_t r = blockIdx.x * 64u + threadIdx.x;
    if (r >= rows) return;
    uint64_t pos = 0;
    const int32_t st = decomp_lane(bits + (uint64_t)r * bytesStride, bytesStride, pos, pc + (uint64_t)r * stride, 1,
                                   (uint32_t)numSamples, (uint32_t)maxSize, mb0, pb, kb);
    numBits[r] = (uint32_t)pos;
    status[r] = st;
}

hipError_t launch_pc_block(const int32_t *in, int32_t *pc, uint32_t rows, uint32_t stride, int32_t num,
                           int16_t *coefs, int32_t numactive, uint32_t chanbits, uint32_t denshift, bool decode,
                           hipStream_t st, bool allowTaps)
{
    if (rows == 0) return hipSuccess;
    // encode direction, 5 taps and more: one chain per half wave, taps across the lanes (alac_stage_taps.hip);
    // fewer taps, or shapes outside that kernel's exact range: one lane per row (option "stage_taps" = 0 forces it)
    const bool noTaps = !allowTaps;
    if (decode) {
        hipLaunchKernelGGL(k_unpc_block, dim3((rows + 63) / 64), dim3(64), 0, st, in, pc, rows, stride, num, coefs,
                           numactive, chanbits, denshift);
    } else if (!noTaps && numactive >= 5 && pc_block_taps_ok(num, numactive, chanbits, denshift)) {
        launch_pc_block_taps(in, pc, rows, stride, num, coefs, numactive, chanbits, denshift, st);
    } else {
        hipLaunchKernelGGL(k_pc_block, dim3((rows + 63) / 64), dim3(64), 0, st, in, pc, rows, stride, num, coefs,
                           numactive, chanbits, denshift);
    }
    return hipGetLastError();
}

hipError_t launch_dyn_comp(uint32_t mb0, uint32_t pb, uint32_t kb, const int32_t *pc, uint32_t rows, uint32_t stride,
                           int32_t numSamples, int32_t bitSize, uint8_t *bits, uint32_t bytesStride,
                           uint32_t *numBits, hipStream_t st)
{
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(k_dyn_comp, dim3((rows + 63) / 64), dim3(64), 0, st, mb0, pb, kb, pc, rows, stride, numSamples,
                       bitSize, (uint32_t *)bits, bytesStride / 4, numBits);
    if (bits) {
        const uint64_t n = (uint64_t)rows * (bytesStride / 4);
        hipLaunchKernelGGL(k_bswap_words, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, (uint32_t *)bits, n);
    }
    return hipGetLastError();
}

hipError_t launch_dyn_decomp(uint32_t mb0, uint32_t pb, uint32_t kb, const uint8_t *bits, uint32_t bytesStride,
                             uint32_t rows, int32_t *pc, uint32_t stride, int32_t numSamples, int32_t maxSize,
                             uint32_t *numBits, int32_t *status, hipStream_t st)
{
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(k_dyn_decomp, dim3((rows + 63) / 64), dim3(64), 0, st, mb0, pb, kb, bits, bytesStride, rows, pc,
                       stride, numSamples, maxSize, numBits, status);
    return hipGetLastError();
}

}  // namespace alacdev
