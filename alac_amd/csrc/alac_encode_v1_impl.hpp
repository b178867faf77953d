// alac_encode_v1_impl.hpp — the tap-parallel encode pipeline: every kernel and the per-depth launcher (included by
// alac_encode_v1_d16/20/24/32.hip, one explicit instantiation each).
//
// Per "packet position" of the segments (position 0 only when every packet is its own segment):
//   k_lms_search1   LPC+mix: the five mixRes passes over N/8 samples walking row 7 (stereo only)
//   k_gol_count     adaptive-Golomb bit counts of those passes, one lane per (chain, pass) stream
//   k_decide1       best mixRes per packet
//   k_lms_search2   LPC+mix: the 8 converge passes of rows 3 and 7 (mono: 7 + 1 passes)
//   k_gol_count     bit counts (first part from the last converge pass, tail from the mixRes = 4 pass:
//                   the stale-predictor-tail quirk of SURVEY.md §3.2)
//   k_decide2       numU / numV, escape estimate, header coefficients
//   k_lms_final     LPC+mix: the final pass over N samples with the chosen row
//   k_gol_final     adaptive-Golomb coder writing the per-channel bit strings, one lane per chain
// then once: k_finalize (sizes + the "too big -> escape" rule), k_scan_sizes, k_pack (alac_encode.hip).
//
// The LPC+mix kernels are the tap-parallel form of alac_lms.hpp: one wave = 8 chains x 8 taps,
// inputs staged in LDS tile by tile, residual tiles flushed to HBM in [sample][stream] order so that
// the lane-per-stream Golomb kernels read them coalesced.
//
// Reference control flow: codec/ALACEncoder.cu:290-558 (EncodeStereo), :812-963 (EncodeMono).
#pragma once
#include <cstdlib>
#include <type_traits>
#include "alac_encode_v1_types.hpp"
#include "alac_lms.hpp"
#include "alac_golomb.hpp"

namespace alacdev {

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N-1 (indices that MUST fold, e.g. into
// register-array subscripts, cannot be left to the unroller)
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

constexpr int kHist = 12;         // samples kept in front of a tile (>= 9: in[j-9] is "top" for 8 taps; a multiple of 4:
                                  // the checked staging works in groups of 4 samples)
// Tile geometry by lanes per chain.  One lane per chain = 64 chains per wave: a 64-step tile would take 23 KB of LDS per
// wave (6 waves per CU); 32 steps take 14.8 KB (10 per CU).  These mappings run where many waves share a SIMD (throughput
// regime) or off the critical path (the 4-tap rows of the search), so the extra tile boundaries are cheap there.
template <int LPC>
struct Geo {
    static constexpr int TILE = LPC == 1 ? 32 : 64;   // predictor steps per LDS tile
    static constexpr int ROWLEN = kHist + TILE;       // samples per row: the history in front of the tile + the tile (76 / 44)
    static constexpr int STRIDE = ROWLEN + 9;         // dwords per input row: + 9 cells of operand prefetch over-read /
                                                      // warm-up parking; odd (89 / 57)
};
constexpr int kZeroCells = 24;    // zeros fed to lanes that hold no active tap


__device__ __forceinline__ bool seg_packet(const SegView &S, uint32_t seg, uint32_t &p, uint32_t &N)
{
    p = 0;
    N = 0;
    if (seg >= S.segEnd) return false;
    const uint32_t p0 = S.segFirst ? S.segFirst[seg] : seg;
    const uint32_t p1 = S.segFirst ? S.segFirst[seg + 1] : seg + 1;
    p = p0 + S.pos;
    // p1 < p0, p1 > numPackets, a segment longer than promised: the table came from the caller unread
    // (alac_hip_encode_segmented) — such a segment has no packets, so no index below is ever out of range
    if (p >= p1 || p1 > S.numPackets || p1 - p0 > S.segMax) {
        p = 0;  // a lane without a packet still forms addresses from p (idle-lane fast paths): keep it in range
        return false;
    }
    N = S.numSamples ? S.numSamples[p] : S.frameSize;
    N = N < S.frameSize ? N : S.frameSize;
    return true;
}

__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// ---- wave roles of the producer / consumer launches ----
// Every wave of these launches is an independent worker (a predictor wave or a coder wave) that wants a SIMD to itself: at
// 10 000 packets the final launch has 938 of them for 1024 SIMDs, and a wave that shares its SIMD runs at ~0.8 of its speed
// and decides when the launch ends.  As single-wave workgroups their SIMD was up to the CU's wave allocator, whose state the
// kernels before leave behind: tools/wave_map.py (round 3) found 14-26 SIMDs with two waves and as many empty whenever an
// unrelated small launch was added or removed upstream, +23 % on the launch.  A workgroup's OWN waves are dealt round the
// CU's four SIMDs, so the workers are launched as workgroups of kWavesPerWg = 4 waves with consecutive worker ids
// (worker = 4 * workgroup + wave): one per SIMD by construction while the launch has at most one workgroup per CU.
constexpr int kWavesPerWg = 4;
struct Worker {
    uint32_t id;   // global worker (wave) index of the launch
    int lane;      // lane inside the wave
    int slot;      // wave inside the workgroup: which LDS block is this worker's
};
__device__ __forceinline__ Worker worker_id()
{
    Worker w;
    w.slot = (int)(threadIdx.x >> 6);
    w.id = blockIdx.x * (uint32_t)kWavesPerWg + (uint32_t)w.slot;
    w.lane = (int)(threadIdx.x & 63);
    return w;
}

// One wave per worker: its LDS operations execute in program order, so phases that hand data through LDS
// only need the compiler not to reorder them (a real fence would also drain the global loads and stores that
// are deliberately left in flight across the tile's compute).
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

// ---- producer -> consumer hand-off inside one launch (cdna_hip_programming.md Guideline 16) ----
// The residual rows are stored with agent-scope atomic stores (global_store ... sc1: written through the
// XCD's L2), so once `s_waitcnt vmcnt(0)` has seen them complete there is nothing left in this L2 for a
// release to write back, and the flag can follow directly.  Measured: an agent-scope release fence here
// (buffer_wbl2 of the WHOLE L2, which also holds the coder waves' dirty bit words) costs ~11 us per
// publish; ALAC_HIP_PUBFENCE=1 puts it back (and publishes every 4 tiles instead of every tile).
__device__ __forceinline__ void publish_rows(uint32_t *flag, uint32_t rows, int lane, bool fence, uint32_t lose = 0)
{
    if (lose) return;  // ALAC_HIP_DEBUG_LOSE_HANDOFF: the consumers must notice
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (fence) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (lane == 0) __hip_atomic_store(flag, rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Consumer: relaxed poll of the (two) producer words, then ONE agent-scope acquire before any newly
// published row is loaded.  The spin is bounded; a consumer that gives up has NOT seen its rows: it raises the
// context's error word (host-mapped, checked at the next synchronize -> the call fails with kALAC_MemFullError) and
// carries on with whatever is there, so the launch still drains.
// Forward progress: producers are the workgroups [0, nLms) of the launch and never wait for anything, so each of them
// finishes in bounded time once it is resident; a consumer only ever waits for producers.  The dispatcher hands out
// workgroups in id order in practice (producers first), but nothing here DEPENDS on that: if consumers were resident
// first they would hold at most their own SIMD slots while spinning with s_sleep, producers fill the remaining slots
// or follow as consumers time out — the outcome is then an error, never a hang and never silent corruption.
struct RowWait {
    const uint32_t *f[4];  // producer progress words (a coder wave's 64 chains come from 1, 2 or 4 predictor waves)
    uint32_t n;
    uint32_t avail, base;
    HandoffCtl ho;
    // producers first .. first + count - 1, of which only those below `limit` exist
    __device__ __forceinline__ void producers(const uint32_t *flags, uint32_t first, uint32_t count, uint32_t limit)
    {
        n = 0;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            f[i] = flags + first;
            if (i < count && first + i < limit) {
                f[i] = flags + first + i;
                n = i + 1;
            }
        }
    }
    __device__ __forceinline__ void operator()(uint32_t rows)
    {
        rows += base;
        if (avail >= rows) return;
        for (uint32_t spins = 0; spins < ho.spinLimit; spins++) {
            uint32_t m = 0xffffffffu;
#pragma unroll
            for (uint32_t i = 0; i < 4; i++)
                if (i < n) m = min(m, __hip_atomic_load(f[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            avail = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
            if (avail >= rows) break;
            __builtin_amdgcn_s_sleep(16);
        }
        if (avail < rows) {
            if (ho.err && (threadIdx.x & 63) == 0) __hip_atomic_store(ho.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            avail = 0xffffffffu;  // stop polling: the call is already lost
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
};



// ================================================================================================
// LPC + mix kernels.  One wave = SLOTS chains (64 / LPC), see alac_lms.hpp for the lane mapping.
// ================================================================================================

// per-wave LDS
template <int LPC>
struct LmsShared {
    static constexpr int SLOTS = 64 / LPC;
    int32_t xs[(SLOTS + 1) * Geo<LPC>::STRIDE];  // chain inputs (mixed / widened samples), one row per chain, overwritten by
                                          // the residuals as the steps pass (+ a dump row for inert lanes)
    int32_t zero[kZeroCells];
    uint32_t pktIdx[SLOTS], pktN[SLOTS];  // per input row: packet and its valid samples
    int32_t rowMix[SLOTS];                // mixRes of the row's packet (this pass)
};

// Staging of x[j0 - kHist .. j0 + kTile) of every chain of the wave, split in two so that the global
// loads of tile t+1 are in flight while tile t computes:
//   stage_load   16-byte PCM loads (4 sample-frames per lane task) into registers
//   stage_store  stereo mix (codec/matrix_enc.cu:72-99 and the 20/24/32-bit forms) or mono widening, then LDS;
//                out-of-range samples -> 0
template <int CH, int LPC>
struct StageRegs {
    static constexpr int SLOTS = 64 / LPC;
    static constexpr int TASKS = (SLOTS / CH) * (Geo<LPC>::ROWLEN / 4);  // (packet, 4-sample group) pairs
    static constexpr int ITERS = (TASKS + 63) / 64;
    int32_t v[ITERS][9];  // raw PCM dwords of each task: 4 sample-frames = CH * bytes-per-sample dwords (<= 8), +1 spare
};

// channel-sample s (0 .. 4 CH - 1) of a task's dwords, after the shift-off of the low bytes — what load_lr /
// load_sample >> SH deliver (codec/matrix_enc.cu:79-82, :129-134, :197-202, :338-342); s is a compile-time
// constant wherever this is called, so everything folds to one or two shifts
template <int DEPTH, int S>
__device__ __forceinline__ int32_t task_sample(const int32_t (&w)[9])
{
    if constexpr (DEPTH == 16) {
        return (S & 1) ? (w[S >> 1] >> 16) : (int32_t)(int16_t)w[S >> 1];
    } else if constexpr (DEPTH == 32) {
        return w[S] >> 16;  // bytesShifted = 2
    } else {
        constexpr int o = 3 * S, i = o >> 2, shb = (o & 3) * 8;
        const uint32_t lo = (uint32_t)w[i], hi = (uint32_t)w[i + 1];
        // the 3 bytes at byte offset o, left aligned in a dword
        uint32_t x;
        if constexpr (shb == 0) x = lo << 8;
        else if constexpr (shb == 8) x = lo & 0xffffff00u;
        else x = ((hi << (32 - shb)) | (lo >> shb)) << 8;
        return DEPTH == 20 ? ((int32_t)x >> 12) : ((int32_t)x >> 16);  // 20: full value; 24: >> 8 (bytesShifted = 1)
    }
}

// vector width of a task's loads: the PCM base is 16-byte aligned, so the packet stride decides whether they apply
template <int DEPTH, int CH>
__device__ __forceinline__ bool task_vec_ok(uint32_t frameBytes)
{
    constexpr int BPF = CH * (int)bytes_per_sample(DEPTH);
    constexpr uint32_t ALIGN = (BPF == 4 || BPF == 8) ? 16 : ((BPF & 1) == 0 ? 8 : 4);
    return (frameBytes & (ALIGN - 1)) == 0;
}

// the BPF dwords of one task (4 sample-frames) starting at src
template <int DEPTH, int CH>
__device__ __forceinline__ void task_load(const uint8_t *src, int32_t (&w)[9])
{
    constexpr int BPF = CH * (int)bytes_per_sample(DEPTH);  // bytes per sample-frame = dwords per 4-frame task
    if constexpr (BPF == 4) {  // 16-bit stereo: the task is 16 bytes, 16-byte aligned
        const int4 w4 = *(const int4 *)src;
        w[0] = w4.x; w[1] = w4.y; w[2] = w4.z; w[3] = w4.w;
    } else if constexpr (BPF == 8) {  // 32-bit stereo: 32 bytes, 16-byte aligned
        const int4 a4 = ((const int4 *)src)[0], b4 = ((const int4 *)src)[1];
        w[0] = a4.x; w[1] = a4.y; w[2] = a4.z; w[3] = a4.w;
        w[4] = b4.x; w[5] = b4.y; w[6] = b4.z; w[7] = b4.w;
    } else if constexpr ((BPF & 1) == 0) {  // 8-byte aligned tasks: 16-bit mono (8 B), 20/24-bit stereo (24 B)
#pragma unroll
        for (int k = 0; k < BPF; k += 2) {
            const int2 w2 = ((const int2 *)src)[k >> 1];
            w[k] = w2.x;
            w[k + 1] = w2.y;
        }
    } else {  // 20/24-bit mono: 12 bytes, dword aligned
#pragma unroll
        for (int k = 0; k < BPF; k++) w[k] = ((const int32_t *)src)[k];
    }
    w[BPF] = 0;  // task_sample's over-read of the last 3-byte sample
}

template <int DEPTH, int CH, int LPC>
__device__ __forceinline__ void stage_load(StageRegs<CH, LPC> &R, const LmsShared<LPC> &sh, const uint8_t *pcm,
                                           uint32_t frameBytes, int j0, int lane)
{
    constexpr int GROUPS = Geo<LPC>::ROWLEN / 4;
    constexpr int BPF = CH * (int)bytes_per_sample(DEPTH);
    const bool vec = task_vec_ok<DEPTH, CH>(frameBytes);
#pragma unroll
    for (int it = 0; it < StageRegs<CH, LPC>::ITERS; it++) {
        const int idx = it * 64 + lane;
        const int q = idx / GROUPS, grp = idx - q * GROUPS;
        const int row = q * CH;
#pragma unroll
        for (int k = 0; k <= BPF; k++) R.v[it][k] = 0;
        if (idx < StageRegs<CH, LPC>::TASKS && vec) {
            const uint32_t N = sh.pktN[row];
            const int jb = j0 - kHist + grp * 4;
            // a group that starts inside the packet is loaded whole: its tail may lie past N but never past the
            // packet's full-size slot in the PCM buffer
            if (jb >= 0 && jb < (int)N)
                task_load<DEPTH, CH>(pcm + (uint64_t)sh.pktIdx[row] * frameBytes + (uint64_t)jb * BPF, R.v[it]);
        }
    }
}

template <int DEPTH, int CH, int LPC>
__device__ __forceinline__ void stage_store(const StageRegs<CH, LPC> &R, LmsShared<LPC> &sh, const uint8_t *pcm,
                                            uint32_t frameBytes, int j0, int lane)
{
    constexpr int GROUPS = Geo<LPC>::ROWLEN / 4;
    constexpr int SH = 8 * (int)bytes_shifted(DEPTH);
    const bool vec = task_vec_ok<DEPTH, CH>(frameBytes);
#pragma unroll
    for (int it = 0; it < StageRegs<CH, LPC>::ITERS; it++) {
        const int idx = it * 64 + lane;
        if (idx >= StageRegs<CH, LPC>::TASKS) continue;
        const int q = idx / GROUPS, grp = idx - q * GROUPS;
        const int row = q * CH;
        const uint32_t N = sh.pktN[row];
        const int jb = j0 - kHist + grp * 4;
        const int mixres = CH == 2 ? sh.rowMix[row] : 0;
        int32_t u[4] = {0, 0, 0, 0}, v[4] = {0, 0, 0, 0};
        if (jb >= 0 && jb < (int)N) {
            if (vec) {
                static_for<4>([&](auto T) {
                    constexpr int t = decltype(T)::value;
                    const bool in = jb + t < (int)N;
                    if constexpr (CH == 2) {
                        const int32_t l = task_sample<DEPTH, 2 * t>(R.v[it]), r = task_sample<DEPTH, 2 * t + 1>(R.v[it]);
                        u[t] = in ? mix_sample(mixres, 0, l, r) : 0;
                        v[t] = in ? mix_sample(mixres, 1, l, r) : 0;
                    } else {
                        u[t] = in ? task_sample<DEPTH, t>(R.v[it]) : 0;
                    }
                });
            } else {
                // packets that are not vector aligned (odd frame sizes): sample by sample
                const uint8_t *pk = pcm + (uint64_t)sh.pktIdx[row] * frameBytes;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (jb + t < (int)N) {
                        if constexpr (CH == 2) {
                            int32_t l, r;
                            load_lr<DEPTH>(pk, (uint32_t)(jb + t), l, r);
                            u[t] = mix_sample(mixres, 0, l, r);
                            v[t] = mix_sample(mixres, 1, l, r);
                        } else {
                            u[t] = load_sample<DEPTH>(pk, (uint32_t)(jb + t)) >> SH;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            sh.xs[row * Geo<LPC>::STRIDE + grp * 4 + t] = u[t];
            if constexpr (CH == 2) sh.xs[(row + 1) * Geo<LPC>::STRIDE + grp * 4 + t] = v[t];
        }
    }
}

// ---- interior fast path of the staging (the tile fully inside every packet of the wave): everything that does not change
// from tile to tile is computed once per pass, and a tile costs a few vector loads plus the mix and the LDS writes per task —
// no bounds checks, no branches.  Only the TILE new samples of a row are loaded: the kHist samples in front of them are the
// last cells of the tile before, which the steps leave intact (residual j lands in cell j - j0, the youngest sample a step
// still needs sits kHist - 9 cells further) — keep_history moves them to the front of the row.  Round 4: until then every
// tile re-read its history from the PCM (48 samples staged per 32 steps): a third more requests than bytes, which reach the
// fabric again because a tile's 3 000+ instructions lie between the two reads of a line, and a third more staging registers —
// the ones k_search1_lane / k_search2_lane spilled.  With TILE * bytes-per-frame = 128 a 16-bit stereo row is one aligned
// cache line per tile.
// (Round 4 measured a COMPACT form of this plan for the 64-chains-per-wave kernels — a 32-bit offset from a wave-uniform base,
// the LDS cell and the mixRes per task, 18 instead of 42 registers — because the compiler spills the 64-bit task addresses there
// (256-register budget).  tools/scratch_sites.py shows why it changed nothing: the spilled values are reloaded in the per-PASS
// set-up, never inside a tile loop; 125 000 packets 9.20 ms compact against 9.13 ms, FETCH_SIZE of k_search1_lane 1.436 GB
// either way.  Not kept.)
template <int CH, int LPC>
struct StagePlan {
    static constexpr int TASKS = (64 / LPC / CH) * (Geo<LPC>::TILE / 4);  // (packet, 4-sample group of the tile) pairs
    static constexpr int ITERS = TASKS / 64;
    static_assert(TASKS % 64 == 0 && ITERS <= StageRegs<CH, LPC>::ITERS, "every lane has a task in every round");
    const uint8_t *pk[ITERS];  // address of the task's 4 sample-frames when the tile starts at j0 = 0
    int xs[ITERS];             // LDS cell of the task's first u (v follows one row further)
    int32_t wl[ITERS], wr[ITERS], vsel[ITERS];  // u = (wl l + wr r) >> 2; v = vsel ? l - r : r
    bool usable;               // dword-aligned packets
};

template <int DEPTH, int CH, int LPC>
__device__ __forceinline__ void stage_plan(StagePlan<CH, LPC> &P, const LmsShared<LPC> &sh, const uint8_t *pcm,
                                           uint32_t frameBytes, int lane)
{
    constexpr int GROUPS = Geo<LPC>::TILE / 4;
    constexpr int BPF = CH * (int)bytes_per_sample(DEPTH);  // bytes per sample-frame = dwords per 4-frame task
    P.usable = task_vec_ok<DEPTH, CH>(frameBytes);
#pragma unroll
    for (int it = 0; it < StagePlan<CH, LPC>::ITERS; it++) {
        const int idx = it * 64 + lane;
        const int q = idx / GROUPS, grp = idx - q * GROUPS;
        const int row = q * CH;
        P.pk[it] = pcm + (uint64_t)sh.pktIdx[row] * frameBytes + grp * (4 * BPF);
        P.xs[it] = row * Geo<LPC>::STRIDE + kHist + grp * 4;
        const int32_t r = CH == 2 ? sh.rowMix[row] : 0;
        P.wl[it] = r ? r : (1 << kMixBits);
        P.wr[it] = r ? (1 << kMixBits) - r : 0;
        P.vsel[it] = r ? -1 : 0;
    }
}

// the TILE new samples of tile j0
template <int DEPTH, int CH, int LPC>
__device__ __forceinline__ void stage_load_fast(StageRegs<CH, LPC> &R, const StagePlan<CH, LPC> &P, int j0)
{
    constexpr int BPF = CH * (int)bytes_per_sample(DEPTH);
    const int64_t byteOff = (int64_t)j0 * BPF;
#pragma unroll
    for (int it = 0; it < StagePlan<CH, LPC>::ITERS; it++) task_load<DEPTH, CH>(P.pk[it] + byteOff, R.v[it]);
}

// The history of the next tile: the last kHist samples of this one, cells TILE .. TILE + kHist - 1 of every row, move to cells
// 0 .. kHist - 1 (HEAD: the tile that starts at sample 0 has zeros there).  Every lane moves 12 / LPC cells.
template <int LPC, bool HEAD = false>
__device__ __forceinline__ void keep_history(LmsShared<LPC> &sh, int lane)
{
    constexpr int SLOTS = 64 / LPC, N = kHist * SLOTS / 64;
    static_assert(kHist * SLOTS % 64 == 0, "every lane moves the same number of cells");
    int32_t v[N];
#pragma unroll
    for (int it = 0; it < N; it++) {
        const int idx = it * 64 + lane, row = idx % SLOTS, c = idx / SLOTS;
        v[it] = HEAD ? 0 : sh.xs[row * Geo<LPC>::STRIDE + Geo<LPC>::TILE + c];
    }
#pragma unroll
    for (int it = 0; it < N; it++) {
        const int idx = it * 64 + lane, row = idx % SLOTS, c = idx / SLOTS;
        sh.xs[row * Geo<LPC>::STRIDE + c] = v[it];
    }
}

template <int DEPTH, int CH, int LPC>
__device__ __forceinline__ void stage_store_fast(const StageRegs<CH, LPC> &R, const StagePlan<CH, LPC> &P,
                                                 LmsShared<LPC> &sh)
{
#pragma unroll
    for (int it = 0; it < StagePlan<CH, LPC>::ITERS; it++) {
        static_for<4>([&](auto T) {
            constexpr int t = decltype(T)::value;
            if constexpr (CH == 2) {
                const int32_t l = task_sample<DEPTH, 2 * t>(R.v[it]), r = task_sample<DEPTH, 2 * t + 1>(R.v[it]);
                // codec/matrix_enc.cu:72-99 with the mixRes = 0 case folded into the weights (4 l >> 2 == l)
                sh.xs[P.xs[it] + t] = (__mul24(P.wl[it], l) + __mul24(P.wr[it], r)) >> kMixBits;
                sh.xs[P.xs[it] + Geo<LPC>::STRIDE + t] = P.vsel[it] ? l - r : r;
            } else {
                sh.xs[P.xs[it] + t] = task_sample<DEPTH, t>(R.v[it]);
            }
        });
    }
}

// The LDS operands of a step (the sample entering the lane's history window, top, in[j]) depend on nothing
// the recurrence produces: the operands of block i+1 (8 steps) are fetched while block i computes.
struct StepOps {
    int32_t nx[8], tp[8], cu[8];
};

__device__ __forceinline__ void load_ops(StepOps &o, const int32_t *pn, const int32_t *pt, const int32_t *pc)
{
#pragma unroll
    for (int s = 0; s < 8; s++) {
        o.nx[s] = pn[s];
        o.tp[s] = pt[s];
        o.cu[s] = pc[s];
    }
}

template <int T, int LPC, bool MASKED, bool DEAD = (T > 4)>
__device__ __forceinline__ void run_block(int32_t (&a)[T], int32_t (&w)[T], const StepOps &o, const LmsLaneT<T> &L, int jb,
                                          int32_t *resAt, uint32_t chanbits)
{
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const int j = jb + s;
        const int32_t liveMask = MASKED ? (((j >= L.jlo) & (j < L.jhi)) ? -1 : 0) : -1;
        resAt[s] = lms_step<T, LPC, MASKED, DEAD>(a, w, o.tp[s], o.cu[s], liveMask, L, chanbits);
#pragma unroll
        for (int i = T - 1; i > 0; i--) w[i] = w[i - 1];
        w[0] = o.nx[s];
    }
}

// per-lane view of its chain inside the wave's LDS
struct LaneView {
    const int32_t *row;   // xs row of the chain (x[j] at row[kHist + j - j0])
    int32_t *res;         // residual row (dump row for inert lanes)
    const int32_t *zero;
    bool feeds;           // this lane holds at least one active tap
};

// one tile [j0, jEnd) of steps; the history windows are (re)loaded from LDS at the tile start
template <int T, int LPC, bool DEAD = (T > 4)>
__device__ __forceinline__ void run_tile(int32_t (&a)[T], const LaneView &V, const LmsLaneT<T> &L, int j0, int jEnd,
                                         uint32_t chanbits)
{
    const int adv = V.feeds ? 1 : 0;
    // x[j] lives at row[kHist + j - j0].  nx of step j = x[j - T h] enters the window for step j + 1
    const int32_t *pn = V.feeds ? V.row + kHist - T * L.h : V.zero;
    const int32_t *pt = V.feeds ? V.row + kHist - 1 - L.na : V.zero;
    const int32_t *pc = V.row + kHist;
    int32_t w[T];
#pragma unroll
    for (int i = 0; i < T; i++) w[i] = V.feeds ? V.row[kHist - 1 - T * L.h - i] : 0;
    StepOps opA, opB;  // ping-pong: no register copies between blocks
    load_ops(opA, pn, pt, pc);
    auto block = [&](const StepOps &cur, StepOps &nxt, int jb) {
        const int o = jb - j0;
        load_ops(nxt, pn + adv * (o + 8), pt + adv * (o + 8), pc + (o + 8));
        const bool allLive = __all((jb >= L.jlo) & (jb + 8 <= L.jhi));  // wave-uniform
        if (allLive)
            run_block<T, LPC, false, DEAD>(a, w, cur, L, jb, V.res + o, chanbits);
        else
            run_block<T, LPC, true, DEAD>(a, w, cur, L, jb, V.res + o, chanbits);
    };
    for (int jb = j0; jb < jEnd; jb += 16) {
        block(opA, opB, jb);
        if (jb + 8 < jEnd) block(opB, opA, jb + 8);
        else opA = opB;
    }
}

// what one wave does in one LPC kernel
struct ChainJob {
    bool active;
    uint32_t seg, ch, p, N;   // chain identity and its packet
    int na;                   // taps of the row it walks
    int16_t *row;             // the persistent coefficient row (first tap)
};

// A pass = one pc_block call over every chain of the wave: `num` samples adapt the row, residual positions
// j < P go to dst[j * streamStride + stream] when store is set.
// ZZ: residuals leave as their zig-zag image 2|del| - (del < 0) (what the final entropy coder starts from)
// Sink: where the residual tile goes.  NoSink = the HBM plane `dst` (the flush below); any other type is a functor
// sink(j0, row) called once per tile with the lane's own LDS row (row[i] = residual of position j0 + i, i < TILE; only
// meaningful for LPC = 1 where lane == slot): the residuals never leave the CU (k_class_final).
struct NoSink {};
template <int DEPTH, int CH, int LPC, bool WT = false, bool ZZ = false, int T = 4, class Sink = NoSink, bool DEAD = (T > 4)>
__device__ __forceinline__ void lms_pass(LmsShared<LPC> &sh, const V1Args &A, const ChainJob &J, int32_t (&a)[T],
                                         uint32_t num, uint32_t P, bool store, int32_t *dst, uint64_t streamStride,
                                         uint32_t stream, int lane, uint32_t *flag = nullptr, uint32_t flagBase = 0,
                                         StageRegs<CH, LPC> *head = nullptr, int headMode = 0, Sink *sink = nullptr)
{
    // head / headMode: the raw PCM of the first tile is the same for every pass a kernel makes over a packet
    // (only the mix weights change): mode 1 keeps it in *head, mode 2 re-mixes it from there instead of loading
    // it again — the first tile's load is the one load of a pass whose latency nothing hides.
    constexpr int SLOTS = 64 / LPC;
    constexpr uint32_t chanBits = DEPTH - 8 * bytes_shifted(DEPTH) + (CH == 2 ? 1 : 0);
    const uint32_t frameBytes = A.S.frameSize * CH * bytes_per_sample(DEPTH);
    const int slot = lane / LPC;
    LmsLaneT<T> L = make_lane<T, LPC>(lane, J.na, J.active ? (int)num : 0);
    // Lanes without work (pad lanes, escape packets in the final pass) must not drag the wave onto the checked
    // paths: they count as "live", as owning every row and as fully inside their packet.  What they compute and
    // store goes to rows / slots nobody reads.
    const uint32_t idleVal = A.idleFast ? 0xffffffffu : 0u;  // what a lane without work reports to the wave minima
    if (!J.active && A.idleFast) {
        L.jlo = 0;
        L.jhi = 0x7fffffff;
    }
    LaneView V;
    V.feeds = J.active && (T * L.h < J.na);
    V.row = sh.xs + slot * Geo<LPC>::STRIDE;
    V.res = V.feeds ? sh.xs + slot * Geo<LPC>::STRIDE : sh.xs + SLOTS * Geo<LPC>::STRIDE;
    V.zero = sh.zero;
    // the lane that flushes slot fs = lane % SLOTS needs that slot's P / stream / activity
    const int fs = lane % SLOTS;
    const uint32_t fP = (uint32_t)__shfl((int)(J.active ? P : 0), fs * LPC);
    const uint32_t fStream = (uint32_t)__shfl((int)stream, fs * LPC);
    const int fNa = __shfl(J.na, fs * LPC);
    const uint32_t runTo = wave_max(J.active ? (store ? (P > num ? P : num) : num) : 0);
    // flush: the lane's column inside a group of LPC rows, and the first row some lane does NOT own
    const uint32_t half = (uint32_t)(lane / SLOTS);
    const uint32_t voff = half * (uint32_t)streamStride + fStream;
    const bool fAct = __shfl((int)J.active, fs * LPC) != 0;
    const uint32_t fPmin = wave_min_u32(fAct ? fP : idleVal);
    StageRegs<CH, LPC> R;
    StagePlan<CH, LPC> SP;
    stage_plan<DEPTH, CH, LPC>(SP, sh, A.S.pcm, frameBytes, lane);
    // tile [j, j + TILE) inside every packet of the wave -> fast staging of tile j (for j > 0 its history is in the row: the
    // tile before was staged whole, by either path)
    const uint32_t nMinRows = wave_min_u32(J.active ? J.N : idleVal);
    auto interior = [&](int j) { return SP.usable && (uint32_t)(j + Geo<LPC>::TILE) <= nMinRows; };
    if (runTo > 0) {
        if (interior(0)) {  // first tile inside every packet: no bounds checks
            keep_history<LPC, true>(sh, lane);
            if (head && headMode == 2) {
                stage_store_fast<DEPTH, CH, LPC>(*head, SP, sh);
            } else {
                stage_load_fast<DEPTH, CH, LPC>(R, SP, 0);
                if (head && headMode == 1) {
                    // the dwords the fast path loaded, one by one: a copy of the whole struct (its unused cells included)
                    // kept *head in scratch memory for the 20- / 24- / 32-bit instances
                    constexpr int BPF = CH * (int)bytes_per_sample(DEPTH);
#pragma unroll
                    for (int it = 0; it < StagePlan<CH, LPC>::ITERS; it++)
#pragma unroll
                        for (int k = 0; k <= BPF; k++) head->v[it][k] = R.v[it][k];
                }
                stage_store_fast<DEPTH, CH, LPC>(R, SP, sh);
            }
        } else {
            stage_load<DEPTH, CH, LPC>(R, sh, A.S.pcm, frameBytes, 0, lane);
            stage_store<DEPTH, CH, LPC>(R, sh, A.S.pcm, frameBytes, 0, lane);
        }
    }
    for (int j0 = 0; j0 < (int)runTo; j0 += Geo<LPC>::TILE) {
        const bool more = j0 + Geo<LPC>::TILE < (int)runTo;
        const bool fastNext = more && interior(j0 + Geo<LPC>::TILE);
        if (fastNext) stage_load_fast<DEPTH, CH, LPC>(R, SP, j0 + Geo<LPC>::TILE);  // in flight under the tile
        else if (more) stage_load<DEPTH, CH, LPC>(R, sh, A.S.pcm, frameBytes, j0 + Geo<LPC>::TILE, lane);
        lds_order();
        // warm-up positions of pc_block (dp_enc.c:90, :108-112): pc[0] = in[0], pc[j] = sext(in[j] - in[j-1]).  They are
        // formed from x[0 .. na] BEFORE the steps overwrite those cells with residuals and parked in the row's prefetch
        // over-read cells (kRowLen .. kRowLen + 8, never data), then moved to cells 0 .. na once the tile has run.
        if (store && j0 == 0) {
            for (int pos = lane / SLOTS; pos <= fNa; pos += LPC) {
                const int32_t *xr = sh.xs + fs * Geo<LPC>::STRIDE + kHist;
                sh.xs[fs * Geo<LPC>::STRIDE + Geo<LPC>::ROWLEN + pos] = pos == 0 ? xr[0] : sext(xr[pos] - xr[pos - 1], 32 - chanBits);
            }
            lds_order();
        }
        const int jEnd = min(j0 + Geo<LPC>::TILE, (int)((runTo + 7) & ~7u));
        run_tile<T, LPC, DEAD>(a, V, L, j0, jEnd, chanBits);
        lds_order();
        if (store) {
            if (j0 == 0) {
                for (int pos = lane / SLOTS; pos <= fNa; pos += LPC)
                    sh.xs[fs * Geo<LPC>::STRIDE + pos] = sh.xs[fs * Geo<LPC>::STRIDE + Geo<LPC>::ROWLEN + pos];
                lds_order();
            }
            // residual tile -> HBM, [sample][stream]: consecutive lanes = consecutive streams.  WT (fused launches):
            // agent-scope stores, written through so that publish_rows has nothing to write back.
            auto put = [&](int32_t *q, int32_t v) {
                if constexpr (ZZ) v = (int32_t)(((uint32_t)v << 1) ^ (uint32_t)(v >> 31));
                if constexpr (WT) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *q = v;
            };
            if constexpr (!std::is_same<Sink, NoSink>::value) {
                static_assert(LPC == 1, "a tile sink reads the lane's own row");
                (*sink)(j0, sh.xs + fs * Geo<LPC>::STRIDE);
            }
            if (!std::is_same<Sink, NoSink>::value && !dst) {
                // the sink was the only consumer (k_class_final): no residual plane
            } else if ((uint32_t)(j0 + Geo<LPC>::TILE) <= fPmin) {
                // every lane owns every row of the tile: scalar row base + lane column, no predicate, no branch
                int32_t *tileBase = dst + (uint64_t)j0 * streamStride;
#pragma unroll
                for (int it = 0; it < Geo<LPC>::TILE / LPC; it++)
                    put(tileBase + (uint64_t)(it * LPC) * streamStride + voff, sh.xs[fs * Geo<LPC>::STRIDE + it * LPC + (int)half]);
            } else {
#pragma unroll 4
                for (int it = 0; it < Geo<LPC>::TILE / LPC; it++) {
                    const int jj = it * LPC + lane / SLOTS;
                    const uint32_t j = (uint32_t)(j0 + jj);
                    const int32_t v = sh.xs[fs * Geo<LPC>::STRIDE + jj];
                    if (j < fP) put(dst + (uint64_t)j * streamStride + fStream, v);
                }
            }
            // fused final kernel: tell the coder waves how many residual rows are complete (every 4 tiles)
            if (flag && ((((j0 / Geo<LPC>::TILE) & A.pubMask & 0xff) == (A.pubMask & 0xff)) || !more)) publish_rows(flag, flagBase + (uint32_t)min(j0 + Geo<LPC>::TILE, (int)runTo), lane, (A.pubMask >> 31) != 0, A.ho.lose);
        }
        lds_order();
        if (fastNext) {
            keep_history<LPC>(sh, lane);
            lds_order();
            stage_store_fast<DEPTH, CH, LPC>(R, SP, sh);
        } else if (more) {
            stage_store<DEPTH, CH, LPC>(R, sh, A.S.pcm, frameBytes, j0 + Geo<LPC>::TILE, lane);
        }
    }
}

template <int LPC>
__device__ __forceinline__ void lms_setup(LmsShared<LPC> &sh, const ChainJob &J, int mix, int lane)
{
    constexpr int SLOTS = 64 / LPC;
    if (lane < kZeroCells) sh.zero[lane] = 0;
    if (lane % LPC == 0) {
        const int slot = lane / LPC;
        sh.pktIdx[slot] = J.p;
        sh.pktN[slot] = J.active ? J.N : 0;
        sh.rowMix[slot] = mix;
    }
    (void)SLOTS;
    lds_order();
}

// virgin (wave-uniform): the rows have never been written — this is the first packet position of a call whose coefficient
// state lives in the workspace — so the row IS init_coefs (codec/dp_enc.c:49-60: 38, -29, -2 scaled by 2^9 / 16) and the
// k_init_state launch that used to write it is not needed
template <int LPC, int T = 4>
__device__ __forceinline__ void load_row(const ChainJob &J, int32_t (&a)[T], int lane, bool virgin = false)
{
    const int h = lane & (LPC - 1);
#pragma unroll
    for (int i = 0; i < T; i++) {
        const int k = T * h + i;
        const int32_t init = k == 0 ? 1216 : (k == 1 ? -928 : (k == 2 ? -64 : 0));
        a[i] = (J.active && k < J.na) ? (virgin ? init : (int32_t)J.row[k]) : 0;
    }
}

// codec/ALACEncoder.cu:374-380: first minimum of bits1 + bits2 over mixRes 0..4 (what k_decide1 did; every wave that needs the
// packet's mixRes after the search picks it itself: ten loads)
__device__ __forceinline__ int pick_mixres(const V1Args &A, uint32_t seg)
{
    uint32_t best = 0, minb = 1u << 31;
#pragma unroll
    for (uint32_t r = 0; r <= (uint32_t)kMaxRes; r++) {
        const uint32_t tot = A.bits1[r * A.chainsPad + seg * 2] + A.bits1[r * A.chainsPad + seg * 2 + 1];
        if (tot < minb) {
            minb = tot;
            best = r;
        }
    }
    return (int)best;
}

// numU / numV and the escape estimate of one packet (codec/ALACEncoder.cu:438-461, mono :899-915): what k_decide2 computes,
// as a function so that the waves of a final launch that folds the decision in (k_final_fused<.., FOLD>) can evaluate it
// themselves from the cost words of the search
struct PacketDecision {
    uint32_t num[2];
    uint32_t escape;
};
template <int DEPTH, int CH>
__device__ __forceinline__ PacketDecision decide_packet(const V1Args &A, uint32_t seg, uint32_t N)
{
    constexpr uint32_t SHB = bytes_shifted(DEPTH);
    PacketDecision D;
    D.num[1] = 0;
    uint32_t minBits = 0;
#pragma unroll
    for (uint32_t c = 0; c < (uint32_t)CH; c++) {
        const uint32_t chain = seg * CH + c;
        const uint32_t c4 = A.cost2[chain], c8 = A.cost2[A.chainsPad + chain];
        D.num[c] = c8 < c4 ? 8 : 4;
        minBits += c8 < c4 ? c8 : c4;
    }
    const uint32_t partial = (N != A.S.frameSize);
    minBits += (CH == 2 ? 64 : 32) + (partial ? 32 : 0) + N * (SHB * 8) * CH;
    const uint32_t escapeBits = N * DEPTH * CH + (partial ? 32 : 0) + 16;
    D.escape = minBits >= escapeBits ? 1u : 0u;
    return D;
}

template <int LPC, int T = 4>
__device__ __forceinline__ void store_row(const ChainJob &J, const int32_t (&a)[T], int lane)
{
    const int h = lane & (LPC - 1);
#pragma unroll
    for (int i = 0; i < T; i++)
        if (J.active && T * h + i < J.na) J.row[T * h + i] = (int16_t)a[i];
}

// ---- k_lms_search1: the five mixRes passes over N/8 samples walking row 7 (codec/ALACEncoder.cu:353-379)
// T taps per lane x L lanes per chain = 8 (alac_lms.hpp "lane mappings"): <4, 2> or <8, 1>
template <int DEPTH, int T, int L>
__device__ __forceinline__ void search1_predictor(LmsShared<L> &sh, const V1Args &A, uint32_t block, int lane, uint32_t *flag)
{
    constexpr int SLOTS = 64 / L;
    ChainJob J;
    const uint32_t chain = A.S.segBegin * 2 + block * SLOTS + lane / L;
    J.seg = chain >> 1;
    J.ch = chain & 1;
    J.active = seg_packet(A.S, J.seg, J.p, J.N);
    J.na = 8;
    J.row = A.state + (uint64_t)J.seg * 64 + J.ch * 32 + 16;
    if (A.rowReady && A.S.pos > 0) {
        // Chained batch with overlapped positions: this launch runs BESIDE the final pass of the previous packet position.
        // A chain whose previous packet runs its final pass on the 8-tap row must wait until that pass has stored the row;
        // one that chose 4 taps (or escaped) left the row final when its search ended, before this launch began.
        bool wait = false;
        if (J.active) {
            const PacketRec *prev = A.recs + (J.p - 1);
            wait = !prev->escape && prev->c[J.ch].num == 8;
        }
        bool seen = false;
        for (uint32_t spins = 0; spins < A.ho.spinLimit; spins++) {
            const uint32_t r = wait ? __hip_atomic_load(A.rowReady + chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
            if (__all(r >= A.S.pos)) {
                seen = true;
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
        if (!seen && A.ho.err && lane == 0) __hip_atomic_store(A.ho.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    int32_t a[T];
    load_row<L>(J, a, lane, A.virgin != 0);
    const uint32_t n8 = J.N / 8;
    // The raw PCM of the first tile is the same for all five passes; keeping it in registers spares each pass the one
    // load whose latency nothing hides — worth ~50 registers where a wave has its SIMD to itself.  The 64-chain mapping of
    // the throughput regime would pay for them with its second wave per SIMD and has other waves to hide the load behind.
    constexpr bool keepHead = L != 1;
    StageRegs<2, L> headRegs;
    StageRegs<2, L> *head = keepHead ? &headRegs : nullptr;
    for (int r = 0; r <= kMaxRes; r++) {
        lms_setup<L>(sh, J, r, lane);
        if (flag)
            lms_pass<DEPTH, 2, L, true>(sh, A, J, a, n8, n8, true, A.resA, 5ull * A.chainsPad, (uint32_t)r * A.chainsPad + chain,
                                        lane, flag, (uint32_t)r << 16, head, r == 0 ? 1 : 2);
        else
            lms_pass<DEPTH, 2, L>(sh, A, J, a, n8, n8, true, A.resA, 5ull * A.chainsPad, (uint32_t)r * A.chainsPad + chain, lane,
                                  nullptr, 0, head, r == 0 ? 1 : 2);
    }
    store_row<L>(J, a, lane);
    if (flag) publish_rows(flag, 0xffffffffu, lane, (A.pubMask >> 31) != 0, A.ho.lose);
}

template <int DEPTH, int T, int L>
__global__ __launch_bounds__(64, L == 1 ? 2 : 1) void k_lms_search1(V1Args A)
{
    __shared__ LmsShared<L> sh;
    search1_predictor<DEPTH, T, L>(sh, A, blockIdx.x, threadIdx.x, nullptr);
}

// ---- k_lms_search2: converge passes for numUV = 4 (row 3, one lane per chain) and 8 (row 7, two lanes per
// chain) in one launch: the first nb3 workgroups take the 4-tap rows (codec/ALACEncoder.cu:420-431; mono :881-893)
// RS = 0: row 3 (numUV = 4), RS = 1: row 7 (numUV = 8); T taps per lane x LPC lanes per chain
// (round 3 measured the counts as trailing consumers of this launch — 0.231 ms against 0.152 + 0.082 separate — and round 4
// removed that variant)
template <int DEPTH, int CH, int RS, int T, int LPC>
__device__ __forceinline__ void search2_body(LmsShared<LPC> &sh, const V1Args &A, uint32_t block, int lane)
{
    constexpr int SLOTS = 64 / LPC;
    constexpr int rs = RS;
    ChainJob J;
    const uint32_t chain = A.S.segBegin * CH + block * SLOTS + lane / LPC;
    J.seg = chain / CH;
    J.ch = chain % CH;
    J.active = seg_packet(A.S, J.seg, J.p, J.N);
    J.na = rs ? 8 : 4;
    J.row = A.state + (uint64_t)J.seg * 64 + J.ch * 32 + rs * 16;
    int32_t a[T];
    // row 3 is first touched here; row 7 of a stereo packet comes from the mixRes search, of a mono packet from nowhere
    load_row<LPC>(J, a, lane, A.virgin != 0 && (rs == 0 || CH == 1));
    const int best = (CH == 2 && J.active) ? pick_mixres(A, J.seg) : 0;
    if (CH == 2 && rs == 1 && J.active && J.ch == 0 && (lane % LPC) == 0) A.recs[J.p].mixRes = (uint32_t)best;
    lms_setup<LPC>(sh, J, best, lane);
    const uint32_t n8 = J.N / 8, n32 = J.N / 32;
    StageRegs<CH, LPC> head;
    for (int pass = 0; pass < 8; pass++) {
        const bool last = pass == 7;
        const uint32_t num = (CH == 1 && last) ? n8 : n32;  // mono: the last pass runs N/8 (:893)
        uint32_t P = num > (uint32_t)(J.na + 1) ? num : (uint32_t)(J.na + 1);  // positions pc_block writes ...
        P = P < n8 ? P : n8;                                                   // ... that dyn_comp will read
        lms_pass<DEPTH, CH, LPC>(sh, A, J, a, num, P, last, A.resB, 2ull * A.chainsPad, (uint32_t)rs * A.chainsPad + chain,
                                 lane, nullptr, 0, &head, pass == 0 ? 1 : 2);
    }
    store_row<LPC>(J, a, lane);
}

// <T3, L3>: mapping of the 4-tap rows, <T7, L7>: of the 8-tap rows (<4, 1> and <4, 2>; <2, 2> and <2, 4> for tiny batches)
// WAVES: waves per SIMD the register budget must allow (2 in the throughput regime, where latency is hidden by the other
// wave; 1 where a wave has its SIMD to itself and every spill would sit on its serial chain)
template <int DEPTH, int CH, int T3 = 4, int L3 = 1, int T7 = 4, int L7 = 2, int WAVES = 1>
__global__ __launch_bounds__(64, WAVES) void k_lms_search2(V1Args A, uint32_t nb3)
{
    __shared__ union {
        LmsShared<L3> s1;
        LmsShared<L7> s2;
    } sh;
    if (blockIdx.x < nb3)
        search2_body<DEPTH, CH, 0, T3, L3>(sh.s1, A, blockIdx.x, threadIdx.x);
    else
        search2_body<DEPTH, CH, 1, T7, L7>(sh.s2, A, blockIdx.x - nb3, threadIdx.x);
}

// The latency regime's form of the above as a launch of workers (4-wave workgroups, see worker_id): roles dealt A B B — worker
// 3 t is the 4-tap wave t (64 chains, <4, 1>), workers 3 t + 1, 3 t + 2 the 8-tap waves 2 t, 2 t + 1 (32 chains each, <4, 2>):
// nb3 + nb7 <= 1024 waves get a SIMD each by construction, whatever the launches before left in the CUs' wave allocators.
template <int DEPTH, int CH>
__global__ __launch_bounds__(64 * kWavesPerWg, 1) void k_lms_search2_w(V1Args A, uint32_t nb3, uint32_t nb7)
{
    __shared__ union {
        LmsShared<1> s1;
        LmsShared<2> s2;
    } shAll[kWavesPerWg];
    const Worker W = worker_id();
    auto &sh = shAll[W.slot];
    const uint32_t t = W.id / 3, r = W.id % 3;
    if (r == 0) {
        if (t < nb3) search2_body<DEPTH, CH, 0, 4, 1>(sh.s1, A, t, W.lane);
    } else {
        const uint32_t b = 2 * t + r - 1;
        if (b < nb7) search2_body<DEPTH, CH, 1, 4, 2>(sh.s2, A, b, W.lane);
    }
}

// ---- k_lms_final: final pass with the chosen row (codec/ALACEncoder.cu:505-532, :941)
template <int DEPTH, int CH>
__global__ __launch_bounds__(64) void k_lms_final(V1Args A)
{
    __shared__ LmsShared<2> sh;
    const int lane = threadIdx.x;
    ChainJob J;
    const uint32_t chain = A.S.segBegin * CH + blockIdx.x * 32u + lane / 2;
    J.seg = chain / CH;
    J.ch = chain % CH;
    J.active = seg_packet(A.S, J.seg, J.p, J.N);
    J.na = 4;
    int best = 0;
    uint32_t N = J.N;
    if (J.active) {
        const PacketRec *rec = A.recs + J.p;
        J.na = rec->c[J.ch].num;
        best = (int)rec->mixRes;
        if (rec->escape) {  // escape estimate: the final pass does not run (:463); still stage zeros
            J.active = false;
        }
    }
    J.row = A.state + (uint64_t)J.seg * 64 + J.ch * 32 + (J.na == 8 ? 16 : 0);
    int32_t a[4];
    load_row<2>(J, a, lane);
    lms_setup<2>(sh, J, best, lane);
    lms_pass<DEPTH, CH, 2, false, true>(sh, A, J, a, N, N, true, A.resC, A.chainsPad, chain, lane);
    store_row<2>(J, a, lane);
}

// ================================================================================================
// Golomb kernels: one lane per stream, residuals read coalesced from the [sample][stream] planes
// ================================================================================================

// search1 counts: stream = r * chainsPad + chain, n8 symbols each -> bits1[stream]
template <int CH>
__global__ __launch_bounds__(64) void k_gol_count1(V1Args A, uint32_t chanBits)
{
    __shared__ uint32_t recip[17];
    gol_table_init(recip, threadIdx.x);
    __syncthreads();
    const uint32_t chain = A.S.segBegin * CH + blockIdx.x * 64u + threadIdx.x, r = blockIdx.y;
    const uint32_t t = r * A.chainsPad + chain;
    uint32_t p, N;
    const bool active = seg_packet(A.S, chain / CH, p, N);
    const uint32_t n8 = active ? N / 8 : 0;
    const int32_t *plane = A.resA + (uint64_t)r * A.chainsPad;
    const uint64_t stride = 5ull * A.chainsPad;
    GolF g;
    golf_reset(g);
    golf_stream<false>(g, n8, wave_max(n8), chanBits, recip, one_plane(plane, stride, chain));
    if (active) A.bits1[t] = g.bits;
}

// ---- k_search1_fused: k_lms_search1 and k_gol_count1 in one launch.  Workgroups [0, nLms) walk the five
// mixRes passes and publish (pass << 16) + rows; the count waves of pass r follow them through plane r.
// WALK: ONE count wave per 64 chains that walks the five planes in turn behind its L producers, roles dealt P .. P C like the
// final launch: (L + 1) cblocks workers instead of nLms + 5 cblocks — at 10 000 packets 938 instead of 2190, so that no
// predictor wave (the launch's serial chain) ever shares its SIMD with a count wave.
template <int DEPTH, int T, int L, bool WALK = false>
__global__ __launch_bounds__(64 * kWavesPerWg, 1) void k_search1_fused(V1Args A, uint32_t nLms, uint32_t cblocks, uint32_t chanBits)
{
    __shared__ LmsShared<L> shAll[kWavesPerWg];
    __shared__ uint32_t recipAll[kWavesPerWg][20];
    const Worker W = worker_id();
    LmsShared<L> &sh = shAll[W.slot];
    uint32_t *recip = recipAll[W.slot];
    const int lane = W.lane;
    if constexpr (WALK) {
        if (W.id >= (uint32_t)(L + 1) * cblocks) return;
        const uint32_t t = W.id / (L + 1), r0 = W.id % (L + 1);
        if (r0 < (uint32_t)L) {
            const uint32_t pw = (uint32_t)L * t + r0;
            if (pw < nLms) search1_predictor<DEPTH, T, L>(sh, A, pw, lane, A.flags + pw);
            return;
        }
        gol_table_init(recip, lane);
        lds_order();
        const uint32_t w = t;
        const uint32_t chain = A.S.segBegin * 2 + w * 64u + lane;
        uint32_t p, N;
        const bool active = seg_packet(A.S, chain >> 1, p, N);
        const uint32_t n8 = active ? N / 8 : 0;
        const uint64_t stride = 5ull * A.chainsPad;
        RowWait wait;
        wait.producers(A.flags, L * w, L, nLms);
        wait.avail = 0;
        wait.ho = A.ho;
        const uint32_t nMax = wave_max(n8);
#pragma unroll 1
        for (uint32_t r = 0; r <= (uint32_t)kMaxRes; r++) {
            GolF g;
            golf_reset(g);
            wait.base = r << 16;
            golf_stream<false>(g, n8, nMax, chanBits, recip, one_plane(A.resA + (uint64_t)r * A.chainsPad, stride, chain), wait);
            if (active) A.bits1[r * A.chainsPad + chain] = g.bits;
        }
        return;
    }
    if (W.id >= nLms + 5 * cblocks) return;
    if (W.id < nLms) {
        search1_predictor<DEPTH, T, L>(sh, A, W.id, lane, A.flags + W.id);
    } else {
        gol_table_init(recip, lane);
        lds_order();
        const uint32_t idx = W.id - nLms, r = idx / cblocks, w = idx % cblocks;
        const uint32_t chain = A.S.segBegin * 2 + w * 64u + lane;
        const uint32_t t = r * A.chainsPad + chain;
        uint32_t p, N;
        const bool active = seg_packet(A.S, chain >> 1, p, N);
        const uint32_t n8 = active ? N / 8 : 0;
        const int32_t *plane = A.resA + (uint64_t)r * A.chainsPad;
        const uint64_t stride = 5ull * A.chainsPad;
        GolF g;
        golf_reset(g);
        // the 64 chains of this wave come from 64 / (64 / L) = L producer waves
        RowWait wait;
        wait.producers(A.flags, L * w, L, nLms);
        wait.avail = 0;
        wait.base = r << 16;
        wait.ho = A.ho;
        golf_stream<false>(g, n8, wave_max(n8), chanBits, recip, one_plane(plane, stride, chain), wait);
        if (active) A.bits1[t] = g.bits;
    }
}

// search2 counts: stream = rowsel * chainsPad + chain.  Stereo: positions < P2 come from the last converge
// pass (resB), the tail from the mixRes = 4 search pass (resA) — codec/ALACEncoder.cu:433-445.
template <int CH, class Need>
__device__ __forceinline__ void count2_body(const V1Args &A, uint32_t chain, uint32_t rs, uint32_t chanBits, const uint32_t *recip,
                                            Need &&need)
{
    const uint32_t t = rs * A.chainsPad + chain;
    uint32_t p, N;
    const bool active = seg_packet(A.S, chain / CH, p, N);
    const uint32_t n8 = active ? N / 8 : 0, n32 = N / 32, na = rs ? 8 : 4;
    uint32_t P2 = n8;
    if (CH == 2) {
        P2 = n32 > na + 1 ? n32 : na + 1;
        P2 = P2 < n8 ? P2 : n8;
    }
    const uint64_t strideB = 2ull * A.chainsPad, strideA = 5ull * A.chainsPad;
    GolF g;
    golf_reset(g);
    // Rows below P2 come from the last converge pass (resB), the tail from the mixRes = 4 search pass (resA).  P2 is
    // the same for every full packet, so normally the row ADDRESS is picked with scalar selects and one load is
    // issued per row; only waves that mix packet lengths read both planes and pick per lane.
    const int32_t *planeB = A.resB + (uint64_t)rs * A.chainsPad, *planeA = A.resA + (uint64_t)kMaxRes * A.chainsPad;
    const uint32_t nMax = wave_max(n8);
    const uint32_t p2lo = wave_min_u32(active ? P2 : 0xffffffffu), p2hi = wave_max(active ? P2 : 0u);
    if (CH == 1 || p2lo >= p2hi) {
        RowSrc rs2;
        rs2.p0 = planeB;
        rs2.s0 = strideB;
        rs2.p1 = planeA;
        rs2.s1 = strideA;
        rs2.split = CH == 1 ? 0xffffffffu : p2hi;
        rs2.col = chain;
        golf_stream<false>(g, n8, nMax, chanBits, recip, rs2, need);
    } else {
        golf_stream_fn<false>(g, n8, nMax, chanBits, recip, [&](uint32_t j) {
            const int32_t b = (planeB + j * strideB)[chain], a = (planeA + j * strideA)[chain];
            return j < P2 ? b : a;
        }, need);
    }
    if (active) A.cost2[t] = g.bits * 8 + 16 * na;  // :438, :447 / :899
}

template <int CH>
__global__ __launch_bounds__(64) void k_gol_count2(V1Args A, uint32_t chanBits)
{
    __shared__ uint32_t recip[17];
    gol_table_init(recip, threadIdx.x);
    __syncthreads();
    count2_body<CH>(A, A.S.segBegin * CH + blockIdx.x * 64u + threadIdx.x, blockIdx.y, chanBits, recip, NoWait());
}

// k_gol_count2 as a launch of workers: worker w counts row set w / cblocks of chains (w % cblocks) * 64 ...
template <int CH>
__global__ __launch_bounds__(64 * kWavesPerWg, 1) void k_gol_count2_w(V1Args A, uint32_t cblocks, uint32_t chanBits)
{
    __shared__ uint32_t recipAll[kWavesPerWg][20];
    const Worker W = worker_id();
    if (W.id >= 2 * cblocks) return;
    uint32_t *recip = recipAll[W.slot];
    gol_table_init(recip, W.lane);
    lds_order();
    count2_body<CH>(A, A.S.segBegin * CH + (W.id % cblocks) * 64u + (uint32_t)W.lane, W.id / cblocks, chanBits, recip, NoWait());
}

// SetFastMode (stereo): what EncodeStereoFast fixes instead of searching (codec/ALACEncoder.cu:613-618) — mixRes 0, numU = numV = 8,
// no escape estimate (the escape decision comes from the bits written, :705-729 = k_finalize's rule) — and the header
// coefficients = row 7 as it stands before the pass (:655-671)
#ifdef ALAC_V1_COMMON_TU
static __global__ void k_decide_fast(V1Args A)
{
    const uint32_t seg = A.S.segBegin + blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t p, N;
    if (!seg_packet(A.S, seg, p, N)) return;
    PacketRec *rec = A.recs + p;
    for (uint32_t c = 0; c < 2; c++) {
        rec->c[c].num = 8;
        rec->c[c].bits = 0;
        const int16_t *row = A.state + (uint64_t)seg * 64 + c * 32 + 16;
        for (uint32_t k = 0; k < 8; k++) rec->c[c].coefs[k] = row[k];
    }
    rec->mixRes = 0;
    rec->numSamples = N;
    rec->escape = 0;
}
#endif

// numU / numV, escape estimate (codec/ALACEncoder.cu:438-461, mono :899-915), header coefficients
template <int DEPTH, int CH>
__global__ void k_decide2(V1Args A)
{
    const uint32_t seg = A.S.segBegin + blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t p, N;
    if (!seg_packet(A.S, seg, p, N)) return;
    PacketRec *rec = A.recs + p;
    constexpr uint32_t SHB = bytes_shifted(DEPTH);
    const uint32_t partial = (N != A.S.frameSize);
    uint32_t minBits = 0;
    for (uint32_t c = 0; c < (uint32_t)CH; c++) {
        const uint32_t chain = seg * CH + c;
        const uint32_t c4 = A.cost2[chain], c8 = A.cost2[A.chainsPad + chain];
        const uint32_t num = c8 < c4 ? 8 : 4;
        minBits += c8 < c4 ? c8 : c4;
        rec->c[c].num = (uint16_t)num;
        rec->c[c].bits = 0;
        const int16_t *row = A.state + (uint64_t)seg * 64 + c * 32 + (num == 8 ? 16 : 0);
        for (uint32_t k = 0; k < 8; k++) rec->c[c].coefs[k] = k < num ? row[k] : (int16_t)0;  // :477-485
    }
    if (CH == 1) {
        rec->c[1].num = 0;
        rec->c[1].bits = 0;
        rec->mixRes = 0;
    }
    minBits += (CH == 2 ? 64 : 32) + (partial ? 32 : 0) + N * (SHB * 8) * CH;
    const uint32_t escapeBits = N * DEPTH * CH + (partial ? 32 : 0) + 16;
    rec->numSamples = N;
    rec->escape = minBits >= escapeBits ? 1u : 0u;
}

// final entropy coding, one lane per chain (codec/ALACEncoder.cu:515-531, :944-945)
template <int CH>
__global__ __launch_bounds__(64) void k_gol_final(V1Args A, uint32_t chanBits)
{
    __shared__ uint32_t recip[17];
    gol_table_init(recip, threadIdx.x);
    __syncthreads();
    const uint32_t chain = A.S.segBegin * CH + blockIdx.x * 64u + threadIdx.x;
    uint32_t p, N;
    bool active = seg_packet(A.S, chain / CH, p, N);
    const bool have = active;  // the lane's packet exists (its bit-word slot may be scribbled on even if it escapes)
    PacketRec *rec = A.recs + p;
    if (active && rec->escape) active = false;
    const uint32_t c = chain % CH;
    const uint32_t n = active ? N : 0;
    const int32_t *plane = A.resC;
    const uint64_t stride = A.chainsPad;
    GolF g;
    golf_reset(g);
    uint32_t *slot = A.bitWords + (have ? (uint64_t)p * 2 + c : (uint64_t)A.dumpSlot + c) * A.wcap;
    golf_open(g, slot, A.wcap);
    golf_stream<true, true>(g, n, wave_max(n), chanBits, recip, one_plane(plane, stride, chain), NoWait(), A.idleFast != 0);
    golf_flush<true>(g);
    if (active) rec->c[c].bits = golf_written_bits(g, slot);
}

// ---- k_final_fused: the final predictor pass and the final entropy coder in ONE launch.  Workgroups
// [0, nLms) are predictor waves (32 chains each), the rest are coder waves (64 chains each) that follow their
// two producers through the residual plane, 256 samples behind.  Both kinds are single-wave workgroups and
// together (625 + 313 at 10k packets) still fit one per SIMD, so the ~1.0 ms and ~1.3 ms of the two stages
// overlap instead of adding up.
// SPLIT (tiny batches): the coder of 64 chains is TWO waves.  Workgroups [nLms, nLms + nCoder) code residuals [0, splitAt) of
// their chains; workgroups [nLms + nCoder, nLms + 2 nCoder) first walk those residuals keeping only the coder's state
// (mean, zero-run bookkeeping: about half the instructions of coding — the bit count of that pass is never read, so the
// compiler drops everything that only feeds it), then code [splitAt, N) into a second slot.  A run that is open at the
// split is closed by the second wave, which has counted its zeros from the start.  k_splice_split appends the second
// string to the first afterwards.  A chained file is two chains: its packet position is as long as ONE coder chain.
// FOLD: the launch also does what k_decide2 and k_finalize do (two launches and their boundaries less per packet position):
// every wave evaluates numU / numV / the escape estimate of its packets itself from the search's cost words (decide_packet);
// the predictor lanes leave the header fields in the record (tap count, the row's coefficients BEFORE the pass adapts them,
// sample count), the coder lanes — the only ones that know the coded bits — the final escape decision, the total bits and
// the packet's byte size.  No lane reads a record field another lane of the launch writes.
template <int DEPTH, int CH, int T = 4, int L = 2, bool SPLIT = false, bool FOLD = false, bool LAZY = false>
__global__ __launch_bounds__(64 * kWavesPerWg, 1) void k_final_fused(V1Args A, uint32_t nLms, uint32_t chanBits, uint32_t nCoder, uint32_t nWorkers)
{
    static_assert(!(FOLD && SPLIT), "the folded decision is built for the plain two-lane launch");
    __shared__ LmsShared<L> shAll[kWavesPerWg];
    __shared__ uint32_t recipAll[kWavesPerWg][20];
    const Worker W = worker_id();
    if (W.id >= nWorkers) return;
    LmsShared<L> &sh = shAll[W.slot];
    uint32_t *recip = recipAll[W.slot];
    const int lane = W.lane;
    // Roles.  A coder wave issues one scattered store per symbol (64 lanes, 64 lines): four of them on one CU queue up in that
    // CU's memory pipeline (measured: 2.06 M cycles per coder wave against 1.69 M, +22 % on the launch, when the workers were
    // numbered predictors first and a workgroup was four coders).  So the roles are dealt P P C P P C ... (L = 2 predictor
    // waves feed one coder wave; P P P P C for L = 4): worker (L + 1) t + r is predictor L t + r for r < L and coder t for
    // r = L — at most two coder waves per workgroup, predictors beside them.
    // (SPLIT, the tiny-batch form, has three roles and a handful of waves: predictors first, as before.)
    uint32_t wid = W.id;     // role-relative index: predictor wave wid, resp. coder wave wid - nLms
    if constexpr (!SPLIT) {
        const uint32_t t = W.id / (L + 1), r = W.id % (L + 1);
        wid = r < (uint32_t)L ? (uint32_t)L * t + r : nLms + t;
        if (r < (uint32_t)L && wid >= nLms) return;
    }
    // option "debug_waves": where and when every wave of the launch ran (HW_ID, XCC_ID, s_memtime at entry and exit), 8 dwords
    // per workgroup in A.dbg — tools/wave_map.py
    struct WaveStamp {
        uint32_t *p;
        __device__ __forceinline__ WaveStamp(uint32_t *dbg, int lane) : p(dbg ? dbg + 8ull * (blockIdx.x * (uint32_t)kWavesPerWg + (threadIdx.x >> 6)) : nullptr)
        {
            if (p && lane == 0) {
                const uint64_t t = __builtin_amdgcn_s_memtime();
                p[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
                p[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
                p[2] = (uint32_t)t;
                p[3] = (uint32_t)(t >> 32);
            }
        }
        __device__ __forceinline__ void done(int lane)
        {
            if (p && lane == 0) {
                const uint64_t t = __builtin_amdgcn_s_memtime();
                p[4] = (uint32_t)t;
                p[5] = (uint32_t)(t >> 32);
                p[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            }
        }
    } stamp(A.dbg, lane);
    if (wid < nLms) {
        ChainJob J;
        const uint32_t chain = A.S.segBegin * CH + wid * (64u / L) + lane / L;
        J.seg = chain / CH;
        J.ch = chain % CH;
        J.active = seg_packet(A.S, J.seg, J.p, J.N);
        J.na = 4;
        int best = 0;
        const uint32_t N = J.N;
        const bool have = J.active;
        if (J.active) {
            if constexpr (FOLD) {
                const PacketDecision D = decide_packet<DEPTH, CH>(A, J.seg, N);
                J.na = (int)D.num[J.ch];
                best = CH == 2 ? (int)A.recs[J.p].mixRes : 0;  // written by the converge launch (search2_body)
                if (D.escape) J.active = false;
            } else {
                const PacketRec *rec = A.recs + J.p;
                J.na = rec->c[J.ch].num;
                best = (int)rec->mixRes;
                if (rec->escape) J.active = false;
            }
        }
        J.row = A.state + (uint64_t)J.seg * 64 + J.ch * 32 + (J.na == 8 ? 16 : 0);
        int32_t a[T];
        load_row<L>(J, a, lane);
        if constexpr (FOLD) {
            if (have) {
                // header fields (codec/ALACEncoder.cu:477-485: the coefficients go into the header before the final pass)
                PacketRec *rec = A.recs + J.p;
                const int h = lane & (L - 1);
#pragma unroll
                for (int i = 0; i < T; i++)
                    if (T * h + i < 8) rec->c[J.ch].coefs[T * h + i] = (T * h + i < J.na) ? (int16_t)a[i] : (int16_t)0;
                if (h == 0) {
                    rec->c[J.ch].num = (uint16_t)J.na;
                    if (J.ch == 0) {
                        rec->numSamples = N;
                        if (CH == 1) {
                            rec->c[1].num = 0;
                            rec->c[1].bits = 0;
                            rec->mixRes = 0;
                        }
                    }
                }
            }
        }
        lms_setup<L>(sh, J, best, lane);
        uint32_t *flag = A.flagsF + wid;
        lms_pass<DEPTH, CH, L, true, true>(sh, A, J, a, N, N, true, A.resC, A.chainsPad, chain, lane, flag);
        store_row<L>(J, a, lane);
        if (A.rowReady) {
            // the next position's search may be waiting for this row (see search1_predictor)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (J.active && J.na == 8 && lane % L == 0 && !A.ho.lose)
                __hip_atomic_store(A.rowReady + chain, A.S.pos + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        publish_rows(flag, 0xffffffffu, lane, (A.pubMask >> 31) != 0, A.ho.lose);  // nothing more will come (also covers inactive waves)
        stamp.done(lane);
    } else {
        gol_table_init(recip, lane);
        lds_order();
        const uint32_t cb = wid - nLms;
        const bool second = SPLIT && cb >= nCoder;
        const uint32_t w = second ? cb - nCoder : cb;
        const uint32_t chain = A.S.segBegin * CH + w * 64u + lane;
        uint32_t p, N;
        bool active = seg_packet(A.S, chain / CH, p, N);
        const bool have = active;
        PacketRec *rec = A.recs + p;
        PacketDecision D;
        D.num[0] = D.num[1] = D.escape = 0;
        if constexpr (FOLD) {
            if (have) D = decide_packet<DEPTH, CH>(A, chain / CH, N);
            if (D.escape) active = false;
        } else {
            if (active && rec->escape) active = false;
        }
        const uint32_t c = chain % CH;
        const uint32_t n = active ? N : 0;
        const int32_t *plane = A.resC;
        const uint64_t stride = A.chainsPad;
        GolF g;
        golf_reset(g);
        const uint64_t slotIdx = have ? (uint64_t)p * 2 + c : (uint64_t)A.dumpSlot + c;
        RowWait wait;
        wait.producers(A.flagsF, L * w, L, nLms);
        wait.avail = 0;
        wait.base = 0;
        wait.ho = A.ho;
        if constexpr (!SPLIT) {
            uint32_t *slot = A.bitWords + slotIdx * A.wcap;
            golf_open(g, slot, A.wcap);
            golf_stream<true, true, RowWait &, LAZY>(g, n, wave_max(n), chanBits, recip, one_plane(plane, stride, chain), wait,
                                                     A.idleFast != 0);
            golf_flush<true, LAZY>(g);
            const uint32_t bits = active ? golf_written_bits<LAZY>(g, slot) : 0u;
            if (active) rec->c[c].bits = bits;
            if constexpr (FOLD) {
                // k_finalize: packet size + the post-hoc "compressed >= escape -> escape" rule (codec/ALACEncoder.cu:537-543,
                // :952-958).  A stereo packet's channels sit in adjacent lanes (chain = 2 * segment + channel, 64 chains per wave).
                const uint32_t other = CH == 2 ? (uint32_t)__shfl_xor((int)bits, 1) : 0u;
                if (have && c == 0) {
                    constexpr uint32_t SHB = bytes_shifted(DEPTH);
                    const uint32_t partial = (N != A.S.frameSize);
                    const uint32_t escapeBits = N * DEPTH * CH + (partial ? 32 : 0) + 16;
                    bool esc = D.escape != 0;
                    uint32_t body = 0;
                    if (!esc) {
                        body = 12 + 4 + (partial ? 32 : 0) + 16 + N * (SHB * 8) * CH + (16 + 16 * D.num[0] + bits);
                        if (CH == 2) body += 16 + 16 * D.num[1] + other;
                        if (body >= escapeBits) esc = true;
                    }
                    if (esc) body = 12 + 4 + (partial ? 32 : 0) + N * DEPTH * CH;
                    rec->escape = esc ? 1u : 0u;
                    rec->totalBits = 7 + body + 3;
                    A.packetBytes[p] = (7 + body + 3 + 7) / 8;
                }
            }
        } else {
            const uint32_t a = A.splitAt;
            const uint32_t nA = min(n, a), nB = n - nA;
            if (!second) {
                uint32_t *slot = A.bitWords + slotIdx * A.wcap;
                golf_open(g, slot, A.wcap);
                // a stream that ends inside this part is finished here; the others stay open for the second wave
                golf_stream<true, true>(g, nA, wave_max(nA), chanBits, recip, one_plane(plane, stride, chain), wait,
                                        A.idleFast != 0, n <= a);
                golf_flush<true>(g);
                if (active) rec->c[c].bits = golf_written_bits(g, slot);
            } else {
                // state only: nothing is written, the bit count is discarded
                golf_stream<false, true>(g, nA, wave_max(nA), chanBits, recip, one_plane(plane, stride, chain), wait,
                                         A.idleFast != 0, false);
                g.bits = 0;
                uint32_t *slot = A.bitWordsB + slotIdx * A.wcap;
                golf_open(g, slot, A.wcap);
                wait.base = a;
                golf_stream<true, true>(g, nB, wave_max(nB), chanBits, recip, one_plane(plane + (uint64_t)a * stride, stride, chain),
                                        wait, A.idleFast != 0, true);
                golf_flush<true>(g);
                if (have) A.bitsB[slotIdx] = (active && nB) ? golf_written_bits(g, slot) : 0u;
            }
        }
        stamp.done(lane);
    }
}

// Append the second coder wave's bit string to the first (SPLIT above): one wave per chain slot of the position's packets.
// The first string ends with a left-aligned partial word (golf_flush), the second starts at bit 0 of its slot.
template <int CH>
__global__ __launch_bounds__(64) void k_splice_split(V1Args A)
{
    const uint32_t chain = A.S.segBegin * CH + blockIdx.x;
    uint32_t p, N;
    if (!seg_packet(A.S, chain / CH, p, N)) return;
    PacketRec *rec = A.recs + p;
    if (rec->escape) return;
    const uint32_t c = chain % CH;
    const uint64_t slotIdx = (uint64_t)p * 2 + c;
    const uint32_t lenA = rec->c[c].bits, lenB = A.bitsB[slotIdx];
    if (lenB == 0) return;
    uint32_t *dst = A.bitWords + slotIdx * A.wcap;
    const uint32_t *src = A.bitWordsB + slotIdx * A.wcap;
    const uint32_t first = lenA >> 5, keep = lenA & 31;                 // word the second string starts in, bits of it in use
    const uint32_t last = min((lenA + lenB + 31) >> 5, A.wcap);         // beyond the slot: longer than the escape size anyway
    const uint32_t nwB = (lenB + 31) >> 5;
    for (uint32_t w = first + threadIdx.x; w < last; w += 64) {
        // 32 bits of the second string from bit s (negative only in the first word)
        const int32_t s = (int32_t)(w * 32u - lenA);
        uint32_t v;
        if (s >= 0) {
            const uint32_t i = (uint32_t)s >> 5, sh = (uint32_t)s & 31;
            const uint32_t x = src[i], y = (i + 1 < nwB) ? src[i + 1] : 0u;
            v = sh ? ((x << sh) | (y >> (32 - sh))) : x;
        } else {
            v = src[0] >> (uint32_t)(-s);
        }
        // (bits behind the end of the second string are zeros already: its last word is left-aligned and zero-padded)
        if (w == first && keep) v |= dst[w];
        dst[w] = v;
    }
    if (threadIdx.x == 0) rec->c[c].bits = lenA + lenB;
}


// ================================================================================================
// Throughput regime: the searches with their dyn_comp bit counts IN THE LANE that predicts (one lane per chain, 64 chains per
// wave, CoderSink<false> below).  k_lms_search1<8, 1> -> k_gol_count1 and k_lms_search2<.., 1, .., 1> -> k_gol_count2 give lane i
// of wave w the same chain, so the planes between them (5 + 2 planes: 2.6 GB written and 2.5 GB read per 125 000-packet
// pass) moved every lane's residuals from the lane to itself.  Only the mixRes = 4 pass still leaves its residuals in HBM
// (plane 4 of resA): the numUV decision counts them behind the first P2 residuals of the converge passes — the stale
// predictor tail of codec/ALACEncoder.cu:433-445.
// ================================================================================================
// The lane codes (WRITE) or only counts (!WRITE) each 32-step tile straight from its own LDS row, where the predictor leaves
// the residuals in place (lms_pass): the residuals never leave the CU.
template <bool WRITE, bool LAZY>
struct CoderSink {
    GolF &g;
    uint32_t n;         // this lane's residual count (0: no chain)
    uint32_t nMaxWave;  // longest stream of the wave
    uint32_t nMinWave;  // shortest (lanes without a chain count as 0 unless idleFast)
    uint32_t bitSize;
    const uint32_t *recip;
    __device__ __forceinline__ void operator()(int j0, const int32_t *row)
    {
        constexpr int B = 16;
#pragma unroll 1
        for (int o = 0; o < Geo<1>::TILE; o += B) {
            const uint32_t jb = (uint32_t)(j0 + o);
            if (jb >= nMaxWave) break;  // wave-uniform
            int32_t buf[B];
#pragma unroll
            for (int s = 0; s < B; s++) buf[s] = row[o + s];
            if (jb + B <= nMinWave) {
#pragma unroll
                for (int s = 0; s < B; s++) golf_sym<WRITE, false, false, LAZY>(g, buf[s], true, bitSize, recip);
            } else {
#pragma unroll
                for (int s = 0; s < B; s++) golf_sym<WRITE, true, false, LAZY>(g, buf[s], jb + s < n, bitSize, recip);
            }
        }
    }
};


template <int DEPTH>
__global__ __launch_bounds__(64, 2) void k_search1_lane(V1Args A, uint32_t chanBits)
{
    __shared__ LmsShared<1> sh;
    __shared__ uint32_t recip[17];
    const int lane = threadIdx.x;
    gol_table_init(recip, lane);
    ChainJob J;
    const uint32_t chain = A.S.segBegin * 2 + blockIdx.x * 64u + (uint32_t)lane;
    J.seg = chain >> 1;
    J.ch = chain & 1;
    J.active = seg_packet(A.S, J.seg, J.p, J.N);
    J.na = 8;
    J.row = A.state + (uint64_t)J.seg * 64 + J.ch * 32 + 16;
    int32_t a[8];
    load_row<1, 8>(J, a, lane, A.virgin != 0);
    const uint32_t n8 = J.active ? J.N / 8 : 0;
    const uint32_t nMax = wave_max(n8), nMin = wave_min_u32(n8);  // a lane without residuals puts the wave on the checked path
    for (int r = 0; r <= kMaxRes; r++) {
        lms_setup<1>(sh, J, r, lane);
        GolF g;
        golf_reset(g);
        CoderSink<false, false> sink{g, n8, nMax, nMin, chanBits, recip};
        // only the last pass (mixRes = 4) also leaves its residuals in HBM: k_search2_lane counts its tail
        lms_pass<DEPTH, 2, 1, false, false, 8, CoderSink<false, false>, false>(sh, A, J, a, J.N / 8, J.N / 8, true, r == kMaxRes ? A.resA : nullptr,
                                                                        5ull * A.chainsPad, (uint32_t)r * A.chainsPad + chain, lane,
                                                                        nullptr, 0, nullptr, 0, &sink);
        golf_finish<false>(g, n8 > 0, recip);
        if (J.active) A.bits1[(uint32_t)r * A.chainsPad + chain] = g.bits;
    }
    store_row<1, 8>(J, a, lane);
}

// converge passes of one row set (RS = 0: row 3 / 4 taps, RS = 1: row 7 / 8 taps) + the numUV cost of the row, one lane per chain
template <int DEPTH, int CH, int RS>
__device__ __forceinline__ void search2_lane_body(LmsShared<1> &sh, const uint32_t *recip, const V1Args &A, uint32_t block, int lane,
                                                  uint32_t chanBits)
{
    constexpr int T = RS ? 8 : 4;
    ChainJob J;
    const uint32_t chain = A.S.segBegin * CH + block * 64u + (uint32_t)lane;
    J.seg = chain / CH;
    J.ch = chain % CH;
    J.active = seg_packet(A.S, J.seg, J.p, J.N);
    J.na = T;
    J.row = A.state + (uint64_t)J.seg * 64 + J.ch * 32 + RS * 16;
    int32_t a[T];
    load_row<1, T>(J, a, lane, A.virgin != 0 && (RS == 0 || CH == 1));
    const int best = (CH == 2 && J.active) ? pick_mixres(A, J.seg) : 0;
    if (CH == 2 && RS == 1 && J.active && J.ch == 0) A.recs[J.p].mixRes = (uint32_t)best;
    lms_setup<1>(sh, J, best, lane);
    const uint32_t n8 = J.N / 8, n32 = J.N / 32;
    uint32_t P2 = n8;  // residuals of the last converge pass that the count reads (k_gol_count2)
    if (CH == 2) {
        P2 = n32 > (uint32_t)(T + 1) ? n32 : (uint32_t)(T + 1);
        P2 = P2 < n8 ? P2 : n8;
    }
    if (!J.active) P2 = 0;
    GolF g;
    golf_reset(g);
    for (int pass = 0; pass < 8; pass++) {
        const bool last = pass == 7;
        const uint32_t num = (CH == 1 && last) ? n8 : n32;  // mono: the last pass runs N/8 (:893)
        if (last) {
            CoderSink<false, false> sink{g, P2, wave_max(P2), wave_min_u32(P2), chanBits, recip};
            lms_pass<DEPTH, CH, 1, false, false, T, CoderSink<false, false>, false>(sh, A, J, a, num, P2, true, nullptr, 0, chain, lane, nullptr, 0,
                                                                             nullptr, 0, &sink);
        } else {
            lms_pass<DEPTH, CH, 1, false, false, T, NoSink, false>(sh, A, J, a, num, 0, false, nullptr, 0, chain, lane);
        }
    }
    store_row<1, T>(J, a, lane);
    if (CH == 2) {
        // the tail [P2, n8) comes from the mixRes = 4 search pass (plane 4 of resA), codec/ALACEncoder.cu:433-445
        const int32_t *planeA = A.resA + (uint64_t)kMaxRes * A.chainsPad;
        const uint64_t strideA = 5ull * A.chainsPad;
        const uint32_t nTail = J.active ? n8 - P2 : 0;
        const uint32_t p2lo = wave_min_u32(J.active ? P2 : 0xffffffffu), p2hi = wave_max(J.active ? P2 : 0u);
        if (p2lo >= p2hi) {
            // (idleFast off: a lane whose stream ended with the converge pass must not count rows of the tail)
            golf_stream<false>(g, nTail, wave_max(nTail), chanBits, recip, one_plane(planeA + (uint64_t)p2hi * strideA, strideA, chain),
                               NoWait(), false, false);
        } else {
            golf_stream_fn<false>(g, nTail, wave_max(nTail), chanBits, recip,
                                  [&](uint32_t j) { return (planeA + (uint64_t)(P2 + j) * strideA)[chain]; }, NoWait(), false, false);
        }
    }
    // the stream ends here whether or not it had a tail (a packet of fewer than ~80 samples has none: n8 <= P2) — a run the last
    // counted residual left open is closed now (round 3: fuzz seeds with 17-sample frames caught a finish that only ran for
    // lanes WITH a tail)
    golf_finish<false>(g, P2 > 0, recip);
    if (J.active) A.cost2[(uint32_t)RS * A.chainsPad + chain] = g.bits * 8 + 16 * T;  // :438, :447 / :899
}

template <int DEPTH, int CH>
__global__ __launch_bounds__(64, 2) void k_search2_lane(V1Args A, uint32_t nb3, uint32_t chanBits)
{
    __shared__ LmsShared<1> sh;
    __shared__ uint32_t recip[17];
    gol_table_init(recip, threadIdx.x);
    if (blockIdx.x < nb3)
        search2_lane_body<DEPTH, CH, 0>(sh, recip, A, blockIdx.x, threadIdx.x, chanBits);
    else
        search2_lane_body<DEPTH, CH, 1>(sh, recip, A, blockIdx.x - nb3, threadIdx.x, chanBits);
}

// ================================================================================================
// Final pass by packet class.  After k_decide2 a packet is (a) escaped — nothing left to predict or code, (b) all
// channels on 4 taps, (c) at least one channel on 8 taps.  The v1 final launch ran every chain of the batch on the
// 2-lanes-x-4-taps mapping; here the chains are compacted per class so that each class gets the lane mapping that fits
// it (alac_lms.hpp "lane mappings") and escaped packets cost nothing:
//   throughput regime   8-tap class <8, 1>, 4-tap class <4, 1>: 64 chains per wave, fewest instructions per chain step
//   latency regime      8-tap class <4, 2> / <2, 4>, 4-tap class <4, 1> / <2, 2>: fewer instructions per wave step
// ================================================================================================

// class of packet i of this position: 0 = nothing to do (no packet here, or escaped), else 4 or 8
template <int CH>
__device__ __forceinline__ uint32_t packet_class(const V1Args &A, uint32_t i, uint32_t nseg)
{
    if (i >= nseg) return 0;
    uint32_t p, N;
    if (!seg_packet(A.S, A.S.segBegin + i, p, N)) return 0;
    const PacketRec *rec = A.recs + p;
    if (rec->escape) return 0;
    uint32_t widest = rec->c[0].num;
    if (CH == 2 && rec->c[1].num > widest) widest = rec->c[1].num;
    return widest == 8 ? 8u : 4u;
}

// Deterministic compaction in packet order, two launches of 1024-packet workgroups:
//   k_class_count   per workgroup: packets of each class -> blockCnt[b] = {count8, count4}
//   k_class_assign  per workgroup: its base = sum of the counts before it (every workgroup adds them up itself: a few
//                   hundred words), columns by ballot rank; the last workgroup writes ClassInfo and the pad columns
template <int CH>
__global__ __launch_bounds__(1024) void k_class_count(V1Args A, uint32_t *blockCnt)
{
    __shared__ uint32_t wcnt[2][16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t nseg = A.S.segEnd - A.S.segBegin;
    const uint32_t c = packet_class<CH>(A, blockIdx.x * 1024u + tid, nseg);
    const uint64_t m8 = __ballot(c == 8), m4 = __ballot(c == 4);
    if (lane == 0) {
        wcnt[0][wv] = (uint32_t)__popcll(m8);
        wcnt[1][wv] = (uint32_t)__popcll(m4);
    }
    __syncthreads();
    if (tid < 2) {
        uint32_t t = 0;
        for (uint32_t q = 0; q < 16; q++) t += wcnt[tid][q];
        blockCnt[blockIdx.x * 2 + tid] = t;
    }
}

template <int CH>
__global__ __launch_bounds__(1024) void k_class_assign(V1Args A, const uint32_t *blockCnt)
{
    __shared__ uint32_t wcnt[2][16];
    __shared__ uint32_t part[4][16];  // before-me / total, per class, per wave
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t nseg = A.S.segEnd - A.S.segBegin;
    const uint32_t nblk = gridDim.x, b = blockIdx.x;
    // sums of the per-workgroup counts: before this workgroup, and in all
    uint32_t before8 = 0, before4 = 0, all8 = 0, all4 = 0;
    for (uint32_t q = tid; q < nblk; q += 1024) {
        const uint32_t a8 = blockCnt[2 * q], a4 = blockCnt[2 * q + 1];
        all8 += a8;
        all4 += a4;
        if (q < b) {
            before8 += a8;
            before4 += a4;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        before8 += (uint32_t)__shfl_xor((int)before8, d);
        before4 += (uint32_t)__shfl_xor((int)before4, d);
        all8 += (uint32_t)__shfl_xor((int)all8, d);
        all4 += (uint32_t)__shfl_xor((int)all4, d);
    }
    if (lane == 0) {
        part[0][wv] = before8;
        part[1][wv] = before4;
        part[2][wv] = all8;
        part[3][wv] = all4;
    }
    const uint32_t c = packet_class<CH>(A, b * 1024u + tid, nseg);
    const uint64_t m8 = __ballot(c == 8), m4 = __ballot(c == 4);
    if (lane == 0) {
        wcnt[0][wv] = (uint32_t)__popcll(m8);
        wcnt[1][wv] = (uint32_t)__popcll(m4);
    }
    __syncthreads();
    uint32_t s[4] = {0, 0, 0, 0};
    for (uint32_t q = 0; q < 16; q++) {
        s[0] += part[0][q];
        s[1] += part[1][q];
        s[2] += part[2][q];
        s[3] += part[3][q];
    }
    const uint32_t n8 = s[2] * CH, n4 = s[3] * CH, base4 = (n8 + 63) & ~63u, nCols = (base4 + n4 + 63) & ~63u;
    if (c) {
        const uint32_t k = c == 8 ? 0 : 1;
        uint32_t r = s[k] + (uint32_t)__popcll((k ? m4 : m8) & ((1ull << lane) - 1));
        for (uint32_t q = 0; q < wv; q++) r += wcnt[k][q];
        const uint32_t col = (k ? base4 : 0u) + r * CH;
        const uint32_t chain = (A.S.segBegin + b * 1024u + tid) * CH;
        A.colChain[col] = chain;
        if (CH == 2) A.colChain[col + 1] = chain + 1;
    }
    if (b + 1 == nblk) {
        for (uint32_t q = n8 + tid; q < base4; q += 1024) A.colChain[q] = kNoChain;
        for (uint32_t q = base4 + n4 + tid; q < nCols; q += 1024) A.colChain[q] = kNoChain;
        if (tid == 0) {
            A.cls->n8 = n8;
            A.cls->n4 = n4;
            A.cls->base4 = base4;
            A.cls->nCols = nCols;
        }
    }
}

// the chain of a column -> what lms_pass needs
template <int CH>
__device__ __forceinline__ void class_job(const V1Args &A, uint32_t col, uint32_t limit, ChainJob &J, int &best)
{
    const uint32_t chain = col < limit ? A.colChain[col] : kNoChain;
    J.active = chain != kNoChain;
    J.seg = J.active ? chain / CH : A.S.segBegin;
    J.ch = J.active ? chain % CH : 0;
    J.na = 4;
    best = 0;
    uint32_t p = 0, N = 0;
    const bool have = seg_packet(A.S, J.seg, p, N);
    J.p = p;
    J.N = N;
    if (J.active && have) {
        const PacketRec *rec = A.recs + p;
        J.na = rec->c[J.ch].num;
        best = (int)rec->mixRes;
    } else {
        J.active = false;
        J.N = 0;
    }
    J.row = A.state + (uint64_t)J.seg * 64 + J.ch * 32 + (J.na == 8 ? 16 : 0);
}

// ---- k_class_final: final predictor pass AND final entropy coder of a chain in ONE lane (throughput regime), one launch per
// packet class (region 0 = the 8-tap class, columns [0, base4); region 1 = the 4-tap class, [base4, nCols); the grid is sized
// for the worst case on the host — the class counts live on the device — and surplus workgroups leave at once).
// The lane codes each 32-step tile straight from its own LDS row, where the predictor leaves the residuals in place
// (alac_lms.hpp / lms_pass): no residual plane, no transposed flush, no row loads, and the wave carries two independent serial
// recurrences (sign-LMS and the Golomb mean tracker) instead of one.  (Round 3 also had the predictor and the coder as two
// launches with a 4.1 GB plane between them, and a four-waves-per-workgroup form that gained 0.4 %: both removed in round 4.
// A fused producer/consumer form of this pass lost to k_final_fused in the latency regime, 0.97 vs 0.77 ms: a third hot loop
// body in one launch does not fit the instruction cache the workgroups of a CU share.)
template <int DEPTH, int CH, int T>
__global__ __launch_bounds__(64, 2) void k_class_final(V1Args A, uint32_t chanBits, uint32_t region)
{
    __shared__ LmsShared<1> sh;
    __shared__ uint32_t recip[20];
    const int lane = threadIdx.x;
    const uint32_t wid = blockIdx.x;
    const uint32_t n8 = A.cls->n8, n4 = A.cls->n4, base4 = A.cls->base4, nCols = A.cls->nCols;
    const uint32_t col0 = (region ? base4 : 0u) + wid * 64u;
    if (col0 >= (region ? nCols : base4)) return;
    gol_table_init(recip, lane);
    lds_order();
    ChainJob J;
    int best;
    const uint32_t col = col0 + (uint32_t)lane;
    class_job<CH>(A, col, region ? base4 + n4 : n8, J, best);
    const uint32_t N = J.N;  // 0 for a column without a chain
    PacketRec *rec = A.recs + J.p;
    GolF g;
    golf_reset(g);
    uint32_t *slot = A.bitWords + (J.active ? (uint64_t)J.p * 2 + J.ch : (uint64_t)A.dumpSlot + (lane & 1)) * A.wcap;
    golf_open(g, slot, A.wcap);
    CoderSink<true, true> sink{g, N, wave_max(N), wave_min_u32(N ? N : (A.idleFast ? 0xffffffffu : 0u)), chanBits, recip};
    int32_t a[T];
    load_row<1>(J, a, lane);
    lms_setup<1>(sh, J, best, lane);
    lms_pass<DEPTH, CH, 1, false, false, T, CoderSink<true, true>>(sh, A, J, a, N, N, true, nullptr, 0, col, lane, nullptr, 0, nullptr,
                                                                   0, &sink);
    store_row<1>(J, a, lane);
    golf_finish<true, true>(g, N > 0, recip);
    golf_flush<true, true>(g);
    if (J.active) rec->c[J.ch].bits = golf_written_bits<true>(g, slot);
}

// packet size + the post-hoc "compressed >= escape -> escape" rule (codec/ALACEncoder.cu:537-543, :952-958)
template <int DEPTH, int CH>
__global__ void k_finalize(PacketRec *recs, uint32_t *packetBytes, uint32_t numPackets, uint32_t frameSize, const uint32_t *segBad)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= numPackets) return;
    if (segBad && *segBad) return;  // refused segment table: the records of packets nobody encoded are stale
    PacketRec *rec = recs + p;
    constexpr uint32_t SHB = bytes_shifted(DEPTH);
    const uint32_t N = rec->numSamples;
    const uint32_t partial = (N != frameSize);
    const uint32_t escapeBits = N * DEPTH * CH + (partial ? 32 : 0) + 16;
    uint32_t body = 0;
    bool esc = rec->escape != 0;
    if (!esc) {
        body = 12 + 4 + (partial ? 32 : 0) + 16 + N * (SHB * 8) * CH;
        for (uint32_t c = 0; c < (uint32_t)CH; c++) body += 16 + 16 * rec->c[c].num + rec->c[c].bits;
        if (body >= escapeBits) esc = true;
    }
    if (esc) body = 12 + 4 + (partial ? 32 : 0) + N * DEPTH * CH;
    rec->escape = esc ? 1u : 0u;
    rec->totalBits = 7 + body + 3;
    packetBytes[p] = (7 + body + 3 + 7) / 8;
}


// ================================================================================================
// launcher
// ================================================================================================
// Kernels that do not depend on the bit depth (the Golomb counts and the stagewise final coder, the class layout of the
// throughput regime, the splice of the split coder, SetFastMode's decision) are compiled ONCE, in alac_encode_v1_common.hip,
// which defines these launch wrappers; the four per-depth translation units only call them.  (Round 3 instantiated every one
// of them in each of the four: 48 of the library's kernels were copies.)
void v1c_decide_fast(uint32_t nseg, hipStream_t st, const V1Args &A);
void v1c_gol_count1(int ch, uint32_t cblocks, hipStream_t st, const V1Args &A, uint32_t chanBits);
void v1c_gol_count2(int ch, uint32_t cblocks, hipStream_t st, const V1Args &A, uint32_t chanBits);
void v1c_gol_count2_w(int ch, uint32_t cblocks, hipStream_t st, const V1Args &A, uint32_t chanBits);
void v1c_class_layout(int ch, uint32_t nseg, hipStream_t st, const V1Args &A, uint32_t *blockCnt);
void v1c_splice_split(int ch, uint32_t nseg, hipStream_t st, const V1Args &A);
void v1c_gol_final(int ch, uint32_t cblocks, hipStream_t st, const V1Args &A, uint32_t chanBits);
template <int DEPTH, int CH>
void launch_v1_typed(const V1Args &A0, uint32_t numPackets, uint32_t maxSegPackets, hipStream_t st, hipEvent_t *ev,
                     const PackArgs &pa, const V1Streams &vs, const AlacOptions &opt)
{
    constexpr uint32_t chanBits = DEPTH - 8 * bytes_shifted(DEPTH) + (CH == 2 ? 1 : 0);
    const uint32_t nsegAll = A0.S.numSegments;
    // (Overlapped sub-batches — halves of the batch on two streams — were measured in rounds 1-3 and lost in both regimes,
    // 12.3 / 13.6 ms for 2 / 4 against 11.6 at 125 000 packets; removed in round 4.  The whole batch is one "sub-batch".)
    bool latFoldAny = false;
    {
        V1Args A = A0;
        A.S.segBegin = 0;
        A.S.segEnd = nsegAll;
        uint32_t *blockCnt = (uint32_t *)(A0.cls + 1);
        const uint32_t nseg = nsegAll;
        const uint32_t cblocks = (nseg * CH + 63) / 64;
        hipStream_t sh = st;
        hipEvent_t *evh = ev;  // stage events of the predictor / Golomb stages: block 0 of ev
        // Chained tiny batches (a file = one chain of packets): packet position p + 1's mixRes search only needs the 8-tap
        // rows, which position p leaves alone once its own search is over unless it runs its FINAL pass on them — so the
        // positions alternate between two streams, the search launch of p + 1 starts when decide2 of p has run and its
        // predictor waves wait, per chain, for rows that p's final pass still owns (rowReady).  The rest of p + 1 waits for
        // p's final pass.  58-75 % of packets choose 4 taps on both channels: their successor's search (a third of a
        // position's serial chain) disappears behind the final pass.
        const bool overlap = opt.overlapPos != 0 && !(CH == 2 && opt.fastMode) && CH == 2 && maxSegPackets > 1 && A.narrow != 0 && A.thru == 0 &&
                             A.S.frameSize / 8 < 65536u && opt.fused != 0;
        if (overlap) {
            A.rowReady = A0.ovRowReady;
            A.flagsF = A0.ovFlagsF;
            (void)hipMemsetAsync(A.rowReady, 0, (size_t)A0.chainsPad * 4, sh);
        }
        for (uint32_t pos = 0; pos < maxSegPackets; pos++) {
            A.S.pos = pos;
            hipStream_t sp = (overlap && (pos & 1)) ? vs.side[0] : sh;
            if (overlap && pos > 0) (void)hipStreamWaitEvent(sp, vs.stagger[(pos - 1) & 1], 0);  // decide2 of pos - 1
            hipEvent_t *e = (evh && pos + 1 == maxSegPackets) ? evh : nullptr;
            const bool firstPos = pos == 0;
            if (e) (void)hipEventRecord(e[kStageLms1], sp);
            const bool fused = opt.fused != 0;
            // Two regimes (A.thru, set by launch_encode_v1 from the batch size):
            //  latency     at most ~one predictor wave per SIMD: a stage is as slow as its longest serial chain, so the
            //              predictor and the coder that trails it share ONE launch (producer/consumer through HBM) and a
            //              chain gets two lanes;
            //  throughput  many waves per SIMD: every kernel fills the machine by itself, so the stages run as separate
            //              launches with plain coalesced stores (the 4-byte write-through hand-off stores of the fused
            //              launches are one fabric write each and cap them at ~1 TB/s), a chain's taps sit in one lane
            //              (fewest instructions per chain step), the final pass runs per packet class and the coder
            //              stores only completed words.  125 000 packets: 15.8 -> 10.1 ms.
            const bool thru = A.thru != 0;
            const bool fuse = fused && !thru;
            const uint32_t nLms = (nseg * CH + 31) / 32;
            // Tiny batches (a single chained file, a few hundred files side by side): the chains do not even fill one
            // wave per SIMD at 16 chains per wave, so a chain gets FOUR lanes x 2 taps (two lanes for the 4-tap rows of the
            // search) — ~44 instead of ~62 instructions per wave step on every serial chain of the packet position.
            const bool narrow = A.narrow != 0 && fuse;
            const uint32_t nLms16 = (nseg * CH + 15) / 16;
            A.virgin = firstPos ? A0.virgin : 0u;
            // Latency regime, two lanes per chain: the small launches between the big ones are folded away ("fold" = 0 keeps
            // them): the final launch decides numU / numV / escape and the packet sizes itself (k_final_fused<.., FOLD>), and ONE
            // memset clears the progress words of both producer/consumer launches of the position.
            const bool fast = CH == 2 && opt.fastMode != 0;  // SetFastMode: no search passes at all (mono has no fast form)
            const bool foldOk = fuse && !narrow && !fast;
            const bool latFold = foldOk && opt.fold != 0, oneMemset = latFold;
            latFoldAny = latFoldAny || latFold;
            if (foldOk) {
                A.flags2 = A0.flags + ((A0.chainsPad / 32 + 4) & ~3u);   // behind the (at most chains / 32) words of the search launch
                A.flagsF = A0.ovFlagsF;                                  // the second set
                A.foldDecide = latFold ? 1 : 0;
                if (oneMemset) (void)hipMemsetAsync(A.flags, 0, (size_t)2 * (A0.chainsPad / 8 + 16) * 4, sp);
            }
            if (fast) {
                if (e) {
                    (void)hipEventRecord(e[kStageGol1], sp);
                    (void)hipEventRecord(e[kStageLms2], sp);
                    (void)hipEventRecord(e[kStageGol2], sp);
                }
                v1c_decide_fast(nseg, sp, A);
            } else if constexpr (CH == 2) {
                const uint32_t nLms1 = nLms;
                // the search progress word is (pass << 16) + rows: rows of a pass must stay below 2^16
                if (fuse && narrow && A.S.frameSize / 8 < 65536u) {
                    (void)hipMemsetAsync(A.flags, 0, ((size_t)nLms16 * 4 + 15) & ~(size_t)15, sp);
                    hipLaunchKernelGGL((k_search1_fused<DEPTH, 2, 4>), dim3((nLms16 + 5 * cblocks + kWavesPerWg - 1) / kWavesPerWg), dim3(64 * kWavesPerWg), 0, sp, A, nLms16, cblocks,
                                       chanBits);
                    if (e) (void)hipEventRecord(e[kStageGol1], sp);
                } else if (fuse && A.S.frameSize / 8 < 65536u) {
                    if (!oneMemset) (void)hipMemsetAsync(A.flags, 0, ((size_t)nLms * 4 + 15) & ~(size_t)15, sp);
                    // ONE count wave per 64 chains walks the five planes behind its two producers (WALK)
                    hipLaunchKernelGGL((k_search1_fused<DEPTH, 4, 2, true>), dim3((3 * cblocks + kWavesPerWg - 1) / kWavesPerWg),
                                       dim3(64 * kWavesPerWg), 0, sp, A, nLms1, cblocks, chanBits);
                    if (e) (void)hipEventRecord(e[kStageGol1], sp);
                } else if (thru) {
                    // predictor passes and their bit counts in one lane: no residual planes except the mixRes = 4 pass's
                    hipLaunchKernelGGL((k_search1_lane<DEPTH>), dim3(cblocks), dim3(64), 0, sp, A, chanBits);
                    if (e) (void)hipEventRecord(e[kStageGol1], sp);
                } else {
                    hipLaunchKernelGGL((k_lms_search1<DEPTH, 4, 2>), dim3(nLms1), dim3(64), 0, sp, A);
                    if (e) (void)hipEventRecord(e[kStageGol1], sp);
                    v1c_gol_count1(CH, cblocks, sp, A, chanBits);
                }
            } else if (e) {
                (void)hipEventRecord(e[kStageGol1], sp);
            }
            if (overlap && pos > 0) (void)hipStreamWaitEvent(sp, vs.join[(pos - 1) & 1], 0);  // final pass of pos - 1
            if (!fast) {
                if (e) (void)hipEventRecord(e[kStageLms2], sp);
                const uint32_t nb3 = (nseg * CH + 63) / 64, nb7 = (nseg * CH + 31) / 32;
                if (narrow)
                    hipLaunchKernelGGL((k_lms_search2<DEPTH, CH, 2, 2, 2, 4>), dim3(nb7 + nLms16), dim3(64), 0, sp, A, nb7);
                else if (thru)  // 64 chains per wave for both rows, two waves per SIMD; every lane counts its own residuals
                    hipLaunchKernelGGL((k_search2_lane<DEPTH, CH>), dim3(nb3 + nb3), dim3(64), 0, sp, A, nb3, chanBits);
                else if (fuse)  // latency regime: workers (one wave per SIMD by construction)
                    hipLaunchKernelGGL((k_lms_search2_w<DEPTH, CH>), dim3((3 * nb3 + kWavesPerWg - 1) / kWavesPerWg), dim3(64 * kWavesPerWg), 0, sp, A,
                                       nb3, nb7);
                else
                    hipLaunchKernelGGL((k_lms_search2<DEPTH, CH>), dim3(nb3 + nb7), dim3(64), 0, sp, A, nb3);
                if (e) (void)hipEventRecord(e[kStageGol2], sp);
                if (!thru) {
                    if (fuse)
                        v1c_gol_count2_w(CH, cblocks, sp, A, chanBits);
                    else
                        v1c_gol_count2(CH, cblocks, sp, A, chanBits);
                }
                if (!latFold) hipLaunchKernelGGL((k_decide2<DEPTH, CH>), dim3((nseg + 255) / 256), dim3(256), 0, sp, A);
            }
            if (overlap) (void)hipEventRecord(vs.stagger[pos & 1], sp);
            if (e) (void)hipEventRecord(e[kStageLms3], sp);
            if (thru) {
                // final pass by packet class: compact the packets that still need it (k_class_count, k_class_assign), then per class the
                // lane mapping that fits it — escaped packets cost nothing, all-4-tap packets run 64 chains per wave
                const uint32_t cwaves = (((nseg * CH + 63) & ~63u) + 64) / 64;  // worst case per region, + the padding
                v1c_class_layout(CH, nseg, sp, A, blockCnt);
                // the two classes are independent from here on: predictor -> coder of the 4-tap class on a side stream beside
                // those of the 8-tap class.  Each kernel alone leaves the machine unevenly filled (a few thousand waves of
                // ~1 ms each on 1024 SIMDs, LDS-limited to 6 predictor waves per CU); side by side the light coder waves
                // of one class fill what the predictor waves of the other cannot use.
                hipStream_t s2 = vs.side[0];
                (void)hipEventRecord(vs.fork, sp);
                (void)hipStreamWaitEvent(s2, vs.fork, 0);
                // predictor and coder of a chain in one lane: no residual plane (k_class_final)
                hipLaunchKernelGGL((k_class_final<DEPTH, CH, 8>), dim3(cwaves), dim3(64), 0, sp, A, chanBits, 0u);
                hipLaunchKernelGGL((k_class_final<DEPTH, CH, 4>), dim3(cwaves), dim3(64), 0, s2, A, chanBits, 1u);
                (void)hipEventRecord(vs.join[0], s2);
                (void)hipStreamWaitEvent(sh, vs.join[0], 0);
                if (e) (void)hipEventRecord(e[kStageGol3], sp);
            } else if (narrow) {
                (void)hipMemsetAsync(A.flagsF, 0, ((size_t)nLms16 * 4 + 15) & ~(size_t)15, sp);
                if (A.bitWordsB && A.splitAt >= 48) {
                    hipLaunchKernelGGL((k_final_fused<DEPTH, CH, 2, 4, true>), dim3((nLms16 + 2 * cblocks + kWavesPerWg - 1) / kWavesPerWg), dim3(64 * kWavesPerWg), 0, sp, A, nLms16,
                                       chanBits, cblocks, nLms16 + 2 * cblocks);
                    v1c_splice_split(CH, nseg, sp, A);
                } else {
                    hipLaunchKernelGGL((k_final_fused<DEPTH, CH, 2, 4>), dim3((5 * cblocks + kWavesPerWg - 1) / kWavesPerWg),
                                       dim3(64 * kWavesPerWg), 0, sp, A, nLms16, chanBits, 0u, 5 * cblocks);
                }
                if (e) (void)hipEventRecord(e[kStageGol3], sp);
            } else if (latFold) {
                if (!oneMemset) (void)hipMemsetAsync(A.flagsF, 0, ((size_t)nLms * 4 + 15) & ~(size_t)15, sp);
                // (LAZY = true, the throughput regime's "store completed words only, four at a time", was measured here in round 3
                // with the interleaved roles: coder waves 2.20 M instead of 1.70 M cycles, launch 0.95 instead of 0.75 ms — the
                // queue's selects and branch cost a lone wave more than the scattered stores it saves)
                hipLaunchKernelGGL((k_final_fused<DEPTH, CH, 4, 2, false, true>), dim3((3 * cblocks + kWavesPerWg - 1) / kWavesPerWg),
                                   dim3(64 * kWavesPerWg), 0, sp, A, nLms, chanBits, 0u, 3 * cblocks);
                if (e) (void)hipEventRecord(e[kStageGol3], sp);
            } else if (fuse) {
                (void)hipMemsetAsync(A.flagsF, 0, ((size_t)nLms * 4 + 15) & ~(size_t)15, sp);
                hipLaunchKernelGGL((k_final_fused<DEPTH, CH>), dim3((3 * cblocks + kWavesPerWg - 1) / kWavesPerWg), dim3(64 * kWavesPerWg),
                                   0, sp, A, nLms, chanBits, 0u, 3 * cblocks);
                if (e) (void)hipEventRecord(e[kStageGol3], sp);
            } else {
                hipLaunchKernelGGL((k_lms_final<DEPTH, CH>), dim3((nseg * CH + 31) / 32), dim3(64), 0, sp, A);
                if (e) (void)hipEventRecord(e[kStageGol3], sp);
                v1c_gol_final(CH, cblocks, sp, A, chanBits);
            }
            if (overlap) {
                (void)hipEventRecord(vs.join[pos & 1], sp);
                if (pos + 1 == maxSegPackets && sp != sh) (void)hipStreamWaitEvent(sh, vs.join[pos & 1], 0);
            }
            if (e) (void)hipEventRecord(e[kStageScan], sh);  // end marker of this sub-batch's last stage
        }
    }
    // sizes, scan, pack: once, on the caller's stream; their events live in block 1 of ev
    hipEvent_t *evt = ev ? ev + (size_t)(kNumStages + 1) : nullptr;
    if (evt) (void)hipEventRecord(evt[kStageScan], st);
    if (!latFoldAny)  // (the folded final launches have written the packet sizes)
        hipLaunchKernelGGL((k_finalize<DEPTH, CH>), dim3((numPackets + 255) / 256), dim3(256), 0, st, A0.recs, A0.packetBytes,
                           numPackets, A0.S.frameSize, pa.segBad);
    launch_scan_pack(DEPTH, CH, A0.packetBytes, pa, numPackets, st, evt, false);
}

}  // namespace alacdev
