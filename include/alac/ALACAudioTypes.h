/*
 * ALACAudioTypes.h — the plain-data types of the reference's public interface, re-declared for the
 * MI355X build so that code written against the reference's ALACEncoder/ALACDecoder compiles unchanged.
 * Field order and widths follow codec/ALACAudioTypes.h:136-176 (AudioFormatDescription,
 * ALACSpecificConfig) and :54-75 (status codes, limits) of the reference; BitBuffer follows
 * codec/ALACBitUtilities.h:71-78.
 */
#ifndef ALAC_AMD_AUDIOTYPES_H
#define ALAC_AMD_AUDIOTYPES_H

#include <stdint.h>
#include "ALACBitUtilities.h"

#ifdef __cplusplus
extern "C" {
#endif

#ifndef ALAC_AMD_NOERR_DEFINED
#define ALAC_AMD_NOERR_DEFINED
enum { ALAC_noErr = 0 };
#endif
enum {
    kALAC_UnimplementedError = -4,
    kALAC_FileNotFoundError = -43,
    kALAC_ParamError = -50,
    kALAC_MemFullError = -108
};
enum {
    kALACFormatAppleLossless = 0x616c6163, /* 'alac' */
    kALACFormatLinearPCM = 0x6c70636d      /* 'lpcm' */
};
enum {
    kALACMaxChannels = 8,
    kALACMaxEscapeHeaderBytes = 8,
    kALACMaxSearches = 16,
    kALACMaxCoefs = 16,
    kALACDefaultFramesPerPacket = 4096
};
enum {
    kALACFormatFlagIsFloat = (1 << 0),
    kALACFormatFlagIsBigEndian = (1 << 1),
    kALACFormatFlagIsSignedInteger = (1 << 2),
    kALACFormatFlagIsPacked = (1 << 3),
    kALACFormatFlagIsAlignedHigh = (1 << 4)
};
enum { kALACFormatFlagsNativeEndian = 0 };
enum { kALACVersion = 0, kALACCompatibleVersion = 0, kALACDefaultFrameSize = 4096 };

typedef double alac_float64_t;

typedef struct AudioFormatDescription {
    alac_float64_t mSampleRate;
    uint32_t mFormatID;
    uint32_t mFormatFlags; /* ALAC side: 1..4 = 16/20/24/32-bit source (codec/ALACEncoder.cu:1463-1479) */
    uint32_t mBytesPerPacket;
    uint32_t mFramesPerPacket;
    uint32_t mBytesPerFrame;
    uint32_t mChannelsPerFrame;
    uint32_t mBitsPerChannel;
    uint32_t mReserved;
} AudioFormatDescription;

#pragma pack(push, 1)
typedef struct ALACSpecificConfig {
    uint32_t frameLength;
    uint8_t compatibleVersion;
    uint8_t bitDepth;
    uint8_t pb;
    uint8_t mb;
    uint8_t kb;
    uint8_t numChannels;
    uint16_t maxRun;
    uint32_t maxFrameBytes;
    uint32_t avgBitRate;
    uint32_t sampleRate;
} ALACSpecificConfig;
#pragma pack(pop)

#ifdef __cplusplus
}
#endif
#endif
