/*
 * alac_hip.h — C-ABI of the MI355X-native ALAC hot path (libalac_hip.so).
 *
 * Plain pointers and sizes only.  Every entry point names the reference interface it replaces
 * (paths relative to the reference tree dark-Stallion/alac).  All `d_` pointers are device
 * (HBM) pointers on the context's device; `h_` pointers are host pointers.  Calls are enqueued on
 * the context's HIP stream and are asynchronous unless stated otherwise.  Return value is the
 * reference's int32 status convention: 0 = ALAC_noErr (codec/ALACBitUtilities.h:51-54),
 * -50 = kALAC_ParamError, -108 = kALAC_MemFullError, -4 = kALAC_UnimplementedError
 * (codec/ALACAudioTypes.h:54-60).  HIP runtime failures map to -108 (allocation) or -50.
 */
#ifndef ALAC_HIP_H
#define ALAC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ALAC_HIP_noErr = 0,
    ALAC_HIP_UnimplementedError = -4,
    ALAC_HIP_ParamError = -50,
    ALAC_HIP_MemFullError = -108
};

/* Number of int16 values of persistent encoder state per segment and element: the predictor rows the
 * search touches, [U row 3][U row 7][V row 3][V row 7] x 16 coefficients — the live subset of
 * ALACEncoder::mCoefsU/V (codec/ALACEncoder.h:89-90, rows numUV-1 of codec/ALACEncoder.cu:361,429).
 * Mono and stereo streams have one element; a stream of 3..8 channels has one block per element of its
 * packets (mCoefsU/V[channelIndex] of the element's first channel), laid out [element][segment][64]:
 * alac_hip_state_int16(fmt) = 64 x elements is the per-segment total. */
#define ALAC_HIP_STATE_INT16 64

typedef struct alac_hip_ctx alac_hip_ctx;

/* What InitializeEncoder derives from AudioFormatDescription (codec/ALACEncoder.cu:1457-1479)
 * plus SetFrameSize (codec/ALACEncoder.h:47). */
typedef struct alac_hip_format {
    uint32_t frame_size;   /* sample-frames per packet; kALACDefaultFramesPerPacket = 4096 */
    uint32_t bit_depth;    /* 16, 20, 24 or 32 (mFormatFlags 1..4) */
    uint32_t num_channels; /* 1 (ID_SCE), 2 (ID_CPE) or 3..8: the element sequence of sChannelMaps
                            * (codec/ALACEncoder.cu:97-107), e.g. 6 = SCE CPE CPE SCE */
    uint32_t sample_rate;  /* only carried into the magic cookie */
} alac_hip_format;

/* int16 values of coefficient state per segment for this format (64 x elements per packet). */
uint32_t alac_hip_state_int16(const alac_hip_format *fmt);

/* ---- context ---------------------------------------------------------------------------- */

/* Number of HIP devices visible; < 0 on failure. */
int32_t alac_hip_device_count(void);

/* Create a context bound to `device`.  `stream` is a hipStream_t to enqueue on (NULL = a stream
 * the context creates and owns).  Replaces the implicit default-stream/device-0 use of the fork
 * (codec/ALACEncoder.cu:1494-1512). */
int32_t alac_hip_create(alac_hip_ctx **out_ctx, int32_t device, void *stream);
void alac_hip_destroy(alac_hip_ctx *ctx);
/* Block until everything enqueued on the context's stream has completed
 * (the cudaDeviceSynchronize of codec/ALACEncoder.cu:1448).  Returns kALAC_MemFullError if, in any call since the last
 * synchronize, a consumer wave of an in-launch producer/consumer hand-off gave up waiting (a preempted or lost producer):
 * the outputs of those calls are then invalid (the decoder also marks the packets concerned kALAC_ParamError).  The
 * host-buffer entry points below return the same code themselves. */
int32_t alac_hip_synchronize(alac_hip_ctx *ctx);
/* Code-path switches of ONE context (no reference counterpart: the reference has a single path).  The library picks a
 * regime per call from the batch shape; a caller can pin one.  The ALAC_HIP_<KEY> environment variables only provide the
 * defaults a context is created with.  Keys (value range; -1 = automatic):
 *   "thru" (-1/0/1)        encode: throughput regime — one kernel per stage, a chain's predictor and coder in one lane,
 *                          final pass per packet class; automatic above 65 536 chains
 *   "narrow" (-1/0/1)      encode: four lanes per chain instead of two; automatic (-1): up to 11 264 chains (5 632 stereo packets;
 *                          mono: 10 240), and again from 21 761 to 34 816 chains (mono: 26 112), where the two-lane workers no
 *                          longer have a SIMD each
 *   "fused" (0/1)          encode: predictor || entropy coder as producer/consumer launches (latency and tiny regimes);
 *                          0 = one plain kernel per stage ("stagewise": also what frames above 524 287 samples get)
 *   "fold" (0/1)           latency regime: numU / numV / escape decision and the packet sizes inside the final launch
 *   "split_coder" (0/1)    tiny regime: the final coder of a chain on two waves
 *   "overlap_pos" (0/1)    chained tiny batches: packet position p + 1's search beside position p's final pass
 *   "fast_mode" (0/1)      ALACEncoder::SetFastMode: the search-free stereo path (EncodeStereoFast)
 *   "encoder_lane", "decoder_lane" (0/1)   the first-generation lane-per-chain kernels (a second, structurally different
 *                          implementation kept for differential testing)
 *   "dec_fused" (-1/0/1)   decode: entropy wave + its predictor waves in one launch; automatic up to 65 536 chains (mono: 49 152)
 *   "dec_pair" (0/1)       decode, separate launches, 16- / 20- / 24-bit stereo: the two predictor lanes of a packet un-mix and
 *                          write the PCM
 *   "dec_direct" (0/1/2)   decode, separate launches, 16-bit: the kernels read the caller's stream (dword aligned) instead of a
 *                          staged copy: 0 never, 1 (default) from 80 000 packets on (below that the swap per word on the
 *                          entropy lanes' chain costs more than the copy), 2 whenever the stream allows it
 *   "stage_taps" (0/1)     alac_hip_pc_block: tap-parallel kernel for 5..30 taps
 *   "debug_waves" (0/1)    diagnostics, see alac_hip_debug_waves_offset
 *   "debug_lose_handoff" (0/1)  TEST switch, the one key that invalidates results by design: producers of the in-launch
 *                          hand-offs never publish, so every call fails with kALAC_MemFullError at the next synchronize
 * Every other setting produces the same bytes; only the kernels that run differ.  Unknown key or a value outside the
 * key's range -> kALAC_ParamError. */
int32_t alac_hip_set_option(alac_hip_ctx *ctx, const char *key, int32_t value);
int32_t alac_hip_get_option(alac_hip_ctx *ctx, const char *key, int32_t *value);
/* Diagnostics: with option "debug_waves" = 1 the fused final launch of alac_hip_encode leaves 8 dwords per workgroup at this
 * byte offset of the caller's workspace (HW_ID, XCC_ID, s_memtime at entry, at exit, HW_ID at exit, 0): where and when every
 * wave ran (tools/wave_map.py).  Batches above 4096 chains only (below, the words are the row-ready flags of chained files). */
uint64_t alac_hip_debug_waves_offset(const alac_hip_format *fmt, uint32_t num_packets, uint32_t num_segments);
/* The encode regime this context would pick for a batch of num_segments independent segments of this format (a static
 * string, from the launcher's own predicates): "throughput" (separate launches, 64 chains per wave), "latency"
 * (producer/consumer launches, two lanes per chain), "tiny" (four lanes per chain: chained files), "stagewise" (option
 * "fused" = 0) or "lane" (first-generation kernel).  For reporting. */
const char *alac_hip_encode_regime(alac_hip_ctx *ctx, const alac_hip_format *fmt, uint32_t num_segments);
/* Text of the last HIP/parameter error on this context ("" if none). */
const char *alac_hip_last_error(const alac_hip_ctx *ctx);
/* The stream the context enqueues on (hipStream_t as void*), for event timing by the caller. */
void *alac_hip_stream(const alac_hip_ctx *ctx);

/* ---- batch encode: replaces InitializeSampling + the per-packet Encode loop ----------------
 * (codec/ALACEncoder.cu:1385-1451, :973-1057, :290-558, :749-806, :812-963; the stage calls
 *  pc_block codec/dp_enc.c:77, dyn_comp codec/ag_enc.c:249, mixNN codec/matrix_enc.cu:101-425). */

/* Bytes of device scratch alac_hip_encode needs for this shape. */
uint64_t alac_hip_encode_workspace_bytes(const alac_hip_format *fmt, uint32_t num_packets,
                                         uint32_t num_segments);
/* Upper bound of the packed output of num_packets packets (every packet escaped + header). */
uint64_t alac_hip_encode_max_output_bytes(const alac_hip_format *fmt, uint32_t num_packets);

/*
 * Encode num_packets packets.
 *   d_pcm            packed little-endian interleaved PCM; packet p starts at byte
 *                    p * frame_size * num_channels * bytes_per_sample (convert-utility/main.cu:509)
 *   d_num_samples    [num_packets] sample-frames in each packet (<= frame_size), or NULL = all full
 *                    (the outBytes[] of InitializeSampling, in samples)
 *   d_seg_first      [num_segments + 1] first packet index of each segment (a segment = a run of
 *                    packets chained through the coefficient state), or NULL =
 *                    every packet is its own segment (state = init_coefs, codec/dp_enc.c:49-60)
 *   d_state          [elements][num_segments][ALAC_HIP_STATE_INT16] coefficient rows (elements = 1 up
 *                    to 2 channels); read as the initial state when state_in != 0, always written
 *                    with the final state when non-NULL
 *   d_out            packets written back to back (each byte-aligned, codec/ALACEncoder.cu:1039)
 *   d_packet_bytes   [num_packets] size of each packet (the *ioNumBytes of Encode)
 *   d_packet_offsets [num_packets + 1] exclusive scan of the sizes; last entry = total bytes
 * Returns ParamError for an unsupported format or a too-small workspace/output.
 */
int32_t alac_hip_encode(alac_hip_ctx *ctx, const alac_hip_format *fmt, const void *d_pcm,
                        const uint32_t *d_num_samples, uint32_t num_packets,
                        const uint32_t *d_seg_first, uint32_t num_segments, int16_t *d_state,
                        int32_t state_in, void *d_workspace, uint64_t workspace_bytes,
                        uint8_t *d_out, uint64_t out_capacity, uint32_t *d_packet_bytes,
                        uint64_t *d_packet_offsets);

/* The same with the caller's bound on the packets of a segment.  The pipeline runs once per packet position of the longest
 * segment, so the host has to know that length: alac_hip_encode reads d_seg_first back (one blocking copy per call) when it is
 * given a table; here max_segment_packets (> 0) is taken on trust, nothing is read back and the call stays asynchronous.  The
 * table is checked on the device (ascending, 0 .. num_packets, no segment above the bound); if it fails, the next
 * alac_hip_synchronize returns kALAC_ParamError and NOTHING has been written to d_out (offsets are all zero): every kernel
 * tests the table entry it uses against num_packets and the bound before it forms a packet index (a segment that fails has
 * no packets), and the size scan and the packer produce nothing once the check has failed — an unvalidated table can make
 * the call fail, never make it read or write out of bounds.  An over-estimate only costs idle launches.
 * max_segment_packets = 0: exactly alac_hip_encode.  (The fork's InitializeSampling has no counterpart: one file, one chain.) */
int32_t alac_hip_encode_segmented(alac_hip_ctx *ctx, const alac_hip_format *fmt, const void *d_pcm,
                                  const uint32_t *d_num_samples, uint32_t num_packets,
                                  const uint32_t *d_seg_first, uint32_t num_segments,
                                  uint32_t max_segment_packets, int16_t *d_state, int32_t state_in,
                                  void *d_workspace, uint64_t workspace_bytes, uint8_t *d_out,
                                  uint64_t out_capacity, uint32_t *d_packet_bytes, uint64_t *d_packet_offsets);

/* Per-kernel timing with HIP events recorded on the context's stream around the three kernels of
 * alac_hip_encode (the instrumented counterpart of the dead cudaEvent timing in
 * codec/CudaAlacEncoder.cu:52-65).  begin() arms up to max_calls encode calls; end() synchronises and
 * returns the number of calls timed and the mean milliseconds of every pipeline stage. */
int32_t alac_hip_profile_begin(alac_hip_ctx *ctx, uint32_t max_calls);
/* out_stage_ms: [alac_hip_num_stages()] mean milliseconds of ONE launch of each pipeline stage
 * (alac_hip_stage_name(i)); out_launches: launches of that stage per encode call (the predictor and
 * Golomb stages run once per overlapped sub-batch). */
int32_t alac_hip_profile_end(alac_hip_ctx *ctx, uint32_t *out_calls, float *out_stage_ms,
                             uint32_t *out_launches);
uint32_t alac_hip_num_stages(void);
const char *alac_hip_stage_name(uint32_t stage);

/* 24-byte magic cookie (ALACSpecificConfig, big-endian): GetConfig/GetMagicCookie
 * (codec/ALACEncoder.cu:1082-1140) for <= 2 channels.  Host-only, no device work. */
uint32_t alac_hip_magic_cookie(const alac_hip_format *fmt, uint32_t max_frame_bytes,
                               uint32_t avg_bit_rate, uint8_t *h_cookie24);
/* GetMagicCookieSize / GetMagicCookie for any channel count (codec/ALACEncoder.cu:1097-1140): above 2
 * channels the config is followed by the 12-byte 'chan' atom header and the 12-byte
 * ALACAudioChannelLayout (48 bytes in all; the layout tag in host byte order as the fork writes it).
 * Returns the bytes written, 0 if `capacity` is too small ("no incomplete cookies", :1136-1139). */
uint32_t alac_hip_magic_cookie_size(const alac_hip_format *fmt);
uint32_t alac_hip_magic_cookie_full(const alac_hip_format *fmt, uint32_t max_frame_bytes,
                                    uint32_t avg_bit_rate, uint8_t *h_cookie, uint32_t capacity);

/* ---- batch decode: replaces ALACDecoder::Decode + fillWriteBuffer ---------------------------
 * (codec/ALACDecoder.cu:571-1002, :497-563; dyn_decomp codec/ag_dec.c:272, unpc_block
 *  codec/dp_dec.c:55, gpu_unmixNN codec/ALACDecoder.cu:193-383). */

uint64_t alac_hip_decode_workspace_bytes(const alac_hip_format *fmt, uint32_t num_packets);
/* The same for a stream of known length: packets padded with ID_FIL / ID_DSE elements (which the decoder skips,
 * codec/ALACDecoder.cu:1012-1059) can make a legal stream longer than num_packets regular packets.  alac_hip_decode
 * uses all of the workspace it is given; packets that still do not fit get status kALAC_ParamError. */
uint64_t alac_hip_decode_workspace_bytes_stream(const alac_hip_format *fmt, uint32_t num_packets,
                                                uint64_t stream_bytes);

/*
 * Decode num_packets packets (independent: coefficients travel in each packet header).
 *   h_cookie/size     magic cookie as stored in the CAF 'kuki' chunk (ALACDecoder::Init,
 *                     codec/ALACDecoder.cu:109-190; legacy 'frma'/'alac' wrappers are skipped)
 *   d_stream          packets back to back
 *   d_packet_offsets  [num_packets + 1] byte offset of each packet in d_stream
 *   d_pcm_out         packet p is written at p * frame_size * num_channels * bytes_per_sample
 *   d_num_samples_out [num_packets] decoded sample-frames per packet (outNumSamples of Decode)
 *   d_status          [num_packets] per-packet status (0 or kALAC_ParamError)
 */
int32_t alac_hip_decode(alac_hip_ctx *ctx, const uint8_t *h_cookie, uint32_t cookie_size,
                        const uint8_t *d_stream, const uint64_t *d_packet_offsets,
                        uint32_t num_packets, void *d_workspace, uint64_t workspace_bytes,
                        uint8_t *d_pcm_out, uint32_t *d_num_samples_out, int32_t *d_status);

/* Parse a magic cookie into a format (host only). */
int32_t alac_hip_format_from_cookie(const uint8_t *h_cookie, uint32_t cookie_size,
                                    alac_hip_format *out_fmt);

/* ---- stage-level entry points (device buffers), the extern "C" surface of
 *      codec/dplib.h:49-55, codec/aglib.h:70-74, codec/matrixlib.h:41-60, batched -------------- */

/* pc_block over num_rows independent rows: row r reads d_in + r*row_stride (int32), writes
 * d_pc + r*row_stride (positions < num), adapts d_coefs + r*32 (int16, numactive used) in place
 * (codec/dp_enc.c:77).  row_stride must cover max(num, numactive + 1) readable samples. */
int32_t alac_hip_pc_block(alac_hip_ctx *ctx, const int32_t *d_in, int32_t *d_pc, uint32_t num_rows,
                          uint32_t row_stride, int32_t num, int16_t *d_coefs, int32_t numactive,
                          uint32_t chanbits, uint32_t denshift);
/* unpc_block, same layout (codec/dp_dec.c:55). */
int32_t alac_hip_unpc_block(alac_hip_ctx *ctx, const int32_t *d_pc, int32_t *d_out,
                            uint32_t num_rows, uint32_t row_stride, int32_t num, int16_t *d_coefs,
                            int32_t numactive, uint32_t chanbits, uint32_t denshift);
/* dyn_comp over num_rows rows with AG params (mb0, pb, kb): row r codes num_samples residuals of
 * d_pc + r*row_stride into d_bits + r*bytes_stride starting at bit 0; bit counts to d_num_bits
 * (codec/ag_enc.c:249).  d_bits may be NULL to count only. */
int32_t alac_hip_dyn_comp(alac_hip_ctx *ctx, uint32_t mb0, uint32_t pb, uint32_t kb,
                          const int32_t *d_pc, uint32_t num_rows, uint32_t row_stride,
                          int32_t num_samples, int32_t bit_size, uint8_t *d_bits,
                          uint32_t bytes_stride, uint32_t *d_num_bits);
/* dyn_decomp, inverse layout (codec/ag_dec.c:272); per-row status to d_status. */
int32_t alac_hip_dyn_decomp(alac_hip_ctx *ctx, uint32_t mb0, uint32_t pb, uint32_t kb,
                            const uint8_t *d_bits, uint32_t bytes_stride, uint32_t num_rows,
                            int32_t *d_pc, uint32_t row_stride, int32_t num_samples,
                            int32_t max_size, uint32_t *d_num_bits, int32_t *d_status);

/* ---- host-buffer convenience (synchronous; does its own H2D/D2H and scratch) -----------------
 * What ALACEncoder::Encode / ALACDecoder::Decode callers with host buffers use
 * (convert-utility/main.cu:558, :719). */
int32_t alac_hip_encode_host(alac_hip_ctx *ctx, const alac_hip_format *fmt, const void *h_pcm,
                             uint64_t total_samples, uint32_t segment_packets, int16_t *h_state,
                             int32_t state_in, uint8_t *h_out, uint64_t out_capacity,
                             uint32_t *h_packet_bytes, uint64_t *out_total_bytes);
/* General form: packets back to back at the full-packet stride, packet p holding h_num_samples[p] frames;
 * segment s = packets [h_seg_first[s], h_seg_first[s+1]) chained through the coefficient state (one
 * segment per input file in a multi-file conversion).  h_state: num_segments * 64 int16, may be NULL. */
int32_t alac_hip_encode_host_segments(alac_hip_ctx *ctx, const alac_hip_format *fmt, const void *h_pcm,
                                      const uint32_t *h_num_samples, uint32_t num_packets,
                                      const uint32_t *h_seg_first, uint32_t num_segments, int16_t *h_state,
                                      int32_t state_in, uint8_t *h_out, uint64_t out_capacity,
                                      uint32_t *h_packet_bytes, uint64_t *out_total_bytes);
int32_t alac_hip_decode_host(alac_hip_ctx *ctx, const uint8_t *h_cookie, uint32_t cookie_size,
                             const uint8_t *h_stream, const uint32_t *h_packet_bytes,
                             uint32_t num_packets, uint8_t *h_pcm_out, uint32_t *h_num_samples_out,
                             int32_t *h_status);

/* ---- deterministic synthetic PCM, host side --------------------------------- */
void alac_synth_frame(uint64_t frame_index, uint32_t num_samples, uint32_t bit_depth,
                      uint32_t channels, uint8_t *h_out);
void alac_synth_pcm(uint64_t first_frame, uint32_t num_frames, uint32_t frame_size,
                    uint32_t bit_depth, uint32_t channels, uint8_t *h_out);
/* The same generator on the device (same source, same bytes): frames [first_frame, first_frame + num_frames) of
 * fmt->frame_size sample-frames each, written back to back at d_out on the context's stream (1 or 2 channels).
 * BASELINE.json configs[3]: every rank generates its own shard in HBM. */
int32_t alac_hip_synth_pcm(alac_hip_ctx *ctx, uint64_t first_frame, uint32_t num_frames,
                           const alac_hip_format *fmt, uint8_t *d_out);

/* ---- sharding across GPUs, host only ------------------------------------------------------
 * One process per GPU encodes a contiguous range of the independent units (segments / packets); the shards are byte
 * aligned (codec/ALACEncoder.cu:1039), so the stream is their concatenation in rank order.
 * alac_hip_shard_range: rank `rank` of `world` takes units [*first, *first + *count); the ranges tile [0, num_units).
 * alac_hip_shard_offsets: offsets[r] = byte position of rank r's shard in the re-assembled stream, offsets[world] = its
 * length (the exclusive prefix sum of the all-gathered shard sizes).  Both return 0 or kALAC_ParamError. */
int32_t alac_hip_shard_range(uint64_t num_units, uint32_t world, uint32_t rank, uint64_t *first, uint64_t *count);
int32_t alac_hip_shard_offsets(const uint64_t *shard_bytes, uint32_t world, uint64_t *offsets);

/* ---- stream re-assembly across the GPUs of one node, on RCCL (alac_comm.cpp) ------------------
 * BASELINE north_star: "frames are sharded across the 8 GPUs of one node with RCCL all-gather over xGMI to reassemble the
 * stream".  No reference counterpart (the fork is single-GPU); what makes it legal is the byte alignment of every packet
 * (codec/ALACEncoder.cu:1039).  One process per GPU, one alac_hip_comm per process; librccl is loaded on first use.
 *   alac_hip_comm_unique_id   rank 0 makes the 128-byte ncclUniqueId and hands it to the other ranks by whatever channel
 *                             the launcher has (a file, a socket, torch.distributed's store)
 *   alac_hip_comm_create      ncclCommInitRank on `device`; collective over all `world` ranks.  kALAC_UnimplementedError
 *                             if librccl cannot be loaded
 * Re-assembly of one pass, two phases so that a pipelined caller never waits for the GPU between two encodes; `slot`
 * (0 .. ALAC_HIP_COMM_SLOTS-1) names the pass while both are outstanding:
 *   alac_hip_reassemble_begin   enqueues on `stream` (hipStream_t): all-gather of {shard bytes, shard capacity, output
 *                             capacity} and — when d_packet_bytes is given — of the per-packet sizes into d_all_packet_bytes
 *                             [world * num_packets] (equal num_packets on every rank: the CAF 'pakt' table of the whole
 *                             stream); the table's copy to pinned host memory and an event.  d_shard_bytes: device pointer
 *                             to this rank's byte count (d_packet_offsets + num_packets of alac_hip_encode).  No host wait.
 *   alac_hip_reassemble_finish  waits for that event only, computes the offsets (alac_hip_shard_offsets; copied to
 *                             h_offsets[world + 1] when non-NULL), and enqueues ONE group on `stream`: ncclRecv of every
 *                             peer's shard straight at its offset in d_stream_out, ncclSend of d_shard to every peer, the
 *                             own shard as a device copy.  kALAC_ParamError — on EVERY rank alike, before anything is
 *                             posted — if a shard is longer than its buffer or the stream longer than any rank's
 *                             out_capacity.
 *   alac_hip_reassemble       both phases back to back (slot 0). */
typedef struct alac_hip_comm alac_hip_comm;
#define ALAC_HIP_COMM_ID_BYTES 128
#define ALAC_HIP_COMM_SLOTS 4
int32_t alac_hip_comm_unique_id(uint8_t *h_id);
int32_t alac_hip_comm_create(alac_hip_comm **out_comm, int32_t device, const uint8_t *h_id, uint32_t rank, uint32_t world);
void alac_hip_comm_destroy(alac_hip_comm *comm);
uint32_t alac_hip_comm_rank(const alac_hip_comm *comm);
uint32_t alac_hip_comm_world(const alac_hip_comm *comm);
const char *alac_hip_comm_last_error(const alac_hip_comm *comm);
int32_t alac_hip_reassemble_begin(alac_hip_comm *comm, uint32_t slot, const uint64_t *d_shard_bytes, uint64_t shard_capacity,
                                  uint64_t out_capacity, const uint32_t *d_packet_bytes, uint32_t num_packets,
                                  uint32_t *d_all_packet_bytes, void *stream);
int32_t alac_hip_reassemble_finish(alac_hip_comm *comm, uint32_t slot, const uint8_t *d_shard, uint8_t *d_stream_out,
                                   uint64_t *h_offsets, void *stream);
int32_t alac_hip_reassemble(alac_hip_comm *comm, const uint8_t *d_shard, const uint64_t *d_shard_bytes, uint64_t shard_capacity,
                            const uint32_t *d_packet_bytes, uint32_t num_packets, uint32_t *d_all_packet_bytes,
                            uint8_t *d_stream_out, uint64_t out_capacity, uint64_t *h_offsets, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ALAC_HIP_H */
