/*
 * alacconvert — drop-in for the reference's convert utility (convert-utility/main.cu:73-852) on the MI355X path.
 *
 *   alacconvert <input wav or caf file> <output caf file>        encode (PCM -> ALAC in CAF)
 *   alacconvert <input caf file> <output wav or caf file>        decode (ALAC in CAF -> PCM)
 *
 * Both produce the bytes the reference produces for the same input.  The codec work goes through the same
 * ALACEncoder / ALACDecoder classes the reference's main() uses (include/alac/), which run on the GPU; this file
 * and container.cpp are plain host C++.
 *
 * Extensions (not in the reference):
 *   <output>.m4a / .mp4 on encode, an MP4 / M4A input on decode: ALAC in an ISO base media file (one 'alac' track, the
 *                                             sample description of ALACMagicCookieDescription.txt:177-216) instead of CAF
 *   --batch <in1> <out1> [<in2> <out2> ...]   convert many files in one GPU batch; every output is identical to a
 *                                             single-file run (each file is one independent chain of packets)
 *   --segment-packets K                       encode only: restart the predictor state every K packets so that
 *                                             one long file spreads over the GPU (valid ALAC, NOT byte-identical
 *                                             to the reference's output, about 1 % larger at K = 1)
 *   --devices N                               with --batch: the files are dealt round-robin to N GPUs, one context and
 *                                             one host thread per device (replicas: independent files need no exchange
 *                                             between the GPUs, so there is no RCCL here); outputs are unchanged
 *
 * A single chained file is serial by construction: one file runs as one chain of dependent
 * packets; the GPU pays off with --batch or --segment-packets.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "alac_hip.h"

#include "ALACAudioTypes.h"
#include "ALACDecoder.h"
#include "ALACEncoder.h"
#include "container.h"

using alacfile::Bytes;
using alacfile::InputInfo;

namespace {

struct Job {
    std::string in, out;
    Bytes file;
    InputInfo info;
    Bytes result;
};

void usage()
{
    // main.cu:181-189
    printf("Usage:\n");
    printf("Encode:\n");
    printf("        alacconvert <input wav or caf file> <output caf file>\n");
    printf("Decode:\n");
    printf("        alacconvert <input caf file> <output wav or caf file>\n");
    printf("\n");
    printf("Extensions:\n");
    printf("        alacconvert --batch <in1> <out1> [<in2> <out2> ...]\n");
    printf("        alacconvert --segment-packets K <input wav or caf file> <output caf file>\n");
    printf("        alacconvert --batch --devices N <in1> <out1> [<in2> <out2> ...]\n");
    printf("\n");
}

uint32_t source_bits(uint32_t flag) { return flag == 1 ? 16 : flag == 2 ? 20 : flag == 3 ? 24 : flag == 4 ? 32 : 0; }

AudioFormatDescription alac_format(const InputInfo &in)
{
    // SetOutputFormat, encode branch (main.cu:263-301)
    AudioFormatDescription f;
    memset(&f, 0, sizeof(f));
    f.mFormatID = kALACFormatAppleLossless;
    f.mSampleRate = in.sampleRate;
    f.mFormatFlags = in.bitsPerChannel == 16 ? 1 : in.bitsPerChannel == 20 ? 2 : in.bitsPerChannel == 24 ? 3 : 4;
    f.mFramesPerPacket = kALACDefaultFramesPerPacket;
    f.mChannelsPerFrame = in.channels;
    return f;
}

// ---- encode: all jobs share bit depth and channel count; each file is one segment ----
bool encode_group(std::vector<Job *> &jobs, uint32_t segmentPackets, int device)
{
    const InputInfo &first = jobs[0]->info;
    const uint32_t bps = (first.bitsPerChannel + 7) >> 3, ch = first.channels;  // 20 bits: 3-byte containers (container.cpp)
    const uint32_t bytesPerFrame = bps * ch, frame = kALACDefaultFramesPerPacket;
    const uint64_t packetBytes = (uint64_t)bytesPerFrame * frame;

    ALACEncoder enc;
    enc.SetFrameSize(frame);
    if (device >= 0) enc.SetDevice(device);
    AudioFormatDescription outFmt = alac_format(first);
    if (enc.InitializeEncoder(outFmt, 0) != ALAC_noErr) {
        fprintf(stderr, " Cannot initialise the encoder (status %d)\n", enc.LastStatus());
        return false;
    }

    // packets back to back at the full-packet stride; the reference cuts the payload into full packets plus one
    // partial one (main.cu:476-545) and drops a trailing fraction of a frame (ALACEncoder.cu:984)
    std::vector<uint32_t> numSamples, segFirst(1, 0), firstPacket;
    for (size_t j = 0; j < jobs.size(); j++) {
        const uint64_t n = jobs[j]->info.dataSize;
        const uint64_t full = n / packetBytes, rest = n - full * packetBytes;
        firstPacket.push_back((uint32_t)numSamples.size());
        for (uint64_t p = 0; p < full; p++) numSamples.push_back(frame);
        if (rest) numSamples.push_back((uint32_t)(rest / bytesPerFrame));
        if (segmentPackets == 0) {
            segFirst.push_back((uint32_t)numSamples.size());
        } else {
            for (uint32_t p = firstPacket.back() + segmentPackets; p < numSamples.size(); p += segmentPackets) segFirst.push_back(p);
            segFirst.push_back((uint32_t)numSamples.size());
        }
    }
    // files without payload contribute an empty segment; drop duplicates the segment table cannot hold
    std::vector<uint32_t> segs;
    for (size_t s = 0; s < segFirst.size(); s++)
        if (s == 0 || segFirst[s] != segs.back()) segs.push_back(segFirst[s]);
    const uint32_t np = (uint32_t)numSamples.size();

    Bytes pcm((size_t)np * packetBytes, 0), stream;
    std::vector<uint32_t> sizes(np, 0);
    uint64_t total = 0;
    if (np) {
        for (size_t j = 0; j < jobs.size(); j++) {
            Job &J = *jobs[j];
            uint8_t *dst = pcm.data() + (size_t)firstPacket[j] * packetBytes;
            memcpy(dst, J.file.data() + J.info.dataPos, (size_t)J.info.dataSize);
            if (J.info.bigEndianPcm) alacfile::swap_samples_in_place(dst, J.info.dataSize, J.info.bitsPerChannel);
        }
        stream.resize((size_t)np * (packetBytes + kALACMaxEscapeHeaderBytes));
        const int32_t rc = enc.EncodeSegments(pcm.data(), numSamples.data(), np, segs.data(), (uint32_t)segs.size() - 1,
                                              stream.data(), stream.size(), sizes.data(), &total);
        if (rc != ALAC_noErr) {
            fprintf(stderr, " Encoding failed (status %d)\n", rc);
            return false;
        }
    }
    // per file: cookie + container
    std::vector<uint64_t> offs(np + 1, 0);
    for (uint32_t p = 0; p < np; p++) offs[p + 1] = offs[p] + sizes[p];
    for (size_t j = 0; j < jobs.size(); j++) {
        Job &J = *jobs[j];
        ALACEncoder cookieMaker;  // the cookie carries the file's own sample rate
        cookieMaker.SetFrameSize(frame);
        if (device >= 0) cookieMaker.SetDevice(device);
        AudioFormatDescription f = alac_format(J.info);
        cookieMaker.InitializeEncoder(f, 0);
        uint32_t cookieSize = cookieMaker.GetMagicCookieSize(J.info.channels);
        Bytes cookie(cookieSize, 0);
        cookieMaker.GetMagicCookie(cookie.data(), &cookieSize);
        cookie.resize(cookieSize);
        const uint32_t p0 = firstPacket[j], p1 = j + 1 < jobs.size() ? firstPacket[j + 1] : np;
        alacfile::AlacCafParams cp = {J.info.sampleRate, J.info.channels, J.info.bitsPerChannel, frame, J.info.dataSize};
        std::vector<uint32_t> mine(sizes.begin() + p0, sizes.begin() + p1);
        if (alacfile::has_m4a_extension(J.out)) {
            const alacfile::AlacM4aParams mp = {(uint32_t)J.info.sampleRate, J.info.channels, J.info.bitsPerChannel, frame,
                                                J.info.dataSize / bytesPerFrame};
            J.result = alacfile::build_alac_m4a(mp, cookie, mine, stream.data() + offs[p0], offs[p1] - offs[p0]);
        } else {
            J.result = alacfile::build_alac_caf(cp, cookie, mine, stream.data() + offs[p0], offs[p1] - offs[p0]);
        }
    }
    return true;
}

// ---- decode: jobs with identical cookies decode in one batch ----
bool decode_group(std::vector<Job *> &jobs, const std::vector<alacfile::AlacCafContents> &contents, int device)
{
    const Bytes &cookie = contents[0].cookie;
    ALACDecoder dec;
    if (device >= 0) dec.SetDevice(device);
    Bytes cookieCopy(cookie);
    if (dec.Init(cookieCopy.data(), (uint32_t)cookieCopy.size(), 0) != ALAC_noErr) {
        fprintf(stderr, " Cannot initialise the decoder from the magic cookie\n");
        return false;
    }
    const uint32_t ch = dec.mConfig.numChannels, bits = dec.mConfig.bitDepth, frame = dec.mConfig.frameLength;
    // The 'desc' flag was checked by the caller, but the COOKIE decides what the decoder writes: refuse a cookie whose depth
    // contradicts the file's description.  20 bits: 3-byte samples, left-justified, as the library writes them.
    for (size_t j = 0; j < jobs.size(); j++) {
        if (!(bits == 16 || bits == 20 || bits == 24 || bits == 32) || source_bits(jobs[j]->info.alacSourceFlag) != bits) {
            fprintf(stderr, " Magic cookie bit depth %u does not match the file description: \"%s\"\n", bits, jobs[j]->in.c_str());
            return false;
        }
    }
    const uint32_t bytesPerFrame = ch * ((bits + 7) >> 3);
    std::vector<uint32_t> sizes, firstPacket;
    Bytes stream;
    for (size_t j = 0; j < jobs.size(); j++) {
        firstPacket.push_back((uint32_t)sizes.size());
        uint64_t pos = contents[j].dataPos;
        for (size_t p = 0; p < contents[j].packetBytes.size(); p++) {
            const uint32_t sz = contents[j].packetBytes[p];
            if (!contents[j].packetPos.empty()) pos = contents[j].packetPos[p];  // M4A: chunks need not be contiguous
            stream.insert(stream.end(), jobs[j]->file.begin() + pos, jobs[j]->file.begin() + pos + sz);
            sizes.push_back(sz);
            pos += sz;
        }
    }
    const uint32_t np = (uint32_t)sizes.size();
    Bytes pcm((size_t)np * frame * bytesPerFrame);
    std::vector<uint32_t> ns(np, 0);
    std::vector<int32_t> status(np, 0);
    if (np) {
        const int32_t rc = dec.DecodeBatch(stream.data(), sizes.data(), np, pcm.data(), ns.data(), status.data());
        if (rc != ALAC_noErr) {
            fprintf(stderr, " Decoding failed (status %d)\n", rc);
            return false;
        }
    }
    for (size_t j = 0; j < jobs.size(); j++) {
        Job &J = *jobs[j];
        const uint32_t p0 = firstPacket[j], p1 = j + 1 < jobs.size() ? firstPacket[j + 1] : np;
        Bytes outPcm;
        for (uint32_t p = p0; p < p1; p++) {
            // main.cu:721-724: numFrames of every packet counts, whatever its status
            const uint8_t *src = pcm.data() + (size_t)p * frame * bytesPerFrame;
            outPcm.insert(outPcm.end(), src, src + (size_t)ns[p] * bytesPerFrame);
        }
        if (alacfile::has_wav_extension(J.out)) {
            if (ch > 2) {
                fprintf(stderr, " Cannot decode more than two channels to WAVE\n");  // main.cu:169-174
                return false;
            }
            J.result = alacfile::build_wave(dec.mConfig.sampleRate, ch, bits, outPcm.data(), outPcm.size());
        } else {
            J.result = alacfile::build_pcm_caf(dec.mConfig.sampleRate, ch, bits, outPcm.data(), outPcm.size());
        }
    }
    return true;
}

}  // namespace

int main(int argc, char *argv[])
{
    std::vector<std::string> files;
    bool batch = false, malformed = argc < 2;
    uint32_t segmentPackets = 0, devices = 0;
    for (int i = 1; i < argc && !malformed; i++) {
        const std::string a = argv[i];
        if (a == "-h") {
            malformed = true;
        } else if (a == "--batch") {
            batch = true;
        } else if (a == "--segment-packets" && i + 1 < argc) {
            segmentPackets = (uint32_t)strtoul(argv[++i], nullptr, 10);
            if (segmentPackets == 0) malformed = true;
        } else if (a == "--devices" && i + 1 < argc) {
            devices = (uint32_t)strtoul(argv[++i], nullptr, 10);
            if (devices == 0) malformed = true;
        } else if (!a.empty() && a[0] == '-') {
            printf("unknown option: %s\n", a.c_str());  // main.cu:92-96
            malformed = true;
        } else {
            files.push_back(a);
        }
    }
    if (!malformed && (files.size() < 2 || (files.size() & 1) || (!batch && files.size() != 2))) malformed = true;
    if (!malformed && devices && !batch) malformed = true;  // one file is one serial chain: nothing to deal out
    if (malformed) {
        usage();
        return 1;
    }

    std::vector<Job> jobs(files.size() / 2);
    for (size_t j = 0; j < jobs.size(); j++) {
        Job &J = jobs[j];
        J.in = files[2 * j];
        J.out = files[2 * j + 1];
        if (!alacfile::read_file(J.in, J.file)) {
            fprintf(stderr, " Cannot open file \"%s\"\n", J.in.c_str());
            return 1;
        }
        printf("Input file: %s\n", J.in.c_str());
        printf("Output file: %s\n", J.out.c_str());
        const std::string err = alacfile::sniff_input(J.file, J.info);
        if (!err.empty()) {
            fprintf(stderr, " %s: \"%s\"\n", err.c_str(), J.in.c_str());
            return 1;
        }
        if (!J.info.isAlac) {
            const uint32_t b = J.info.bitsPerChannel;
            if ((b != 16 && b != 20 && b != 24 && b != 32) || J.info.channels < 1 || J.info.channels > 8) {  // kALACMaxChannels
                fprintf(stderr, " File \"%s\'s\" data format is of an unsupported type\n", J.in.c_str());
                return 1;
            }
        }
    }

    // group: encode jobs by (depth, channels); decode jobs by cookie
    std::map<std::string, std::vector<Job *> > groups;
    for (size_t j = 0; j < jobs.size(); j++) {
        Job &J = jobs[j];
        std::string key;
        if (J.info.isAlac) {
            alacfile::AlacCafContents c;
            InputInfo again;
            const std::string err = J.info.kind == alacfile::kM4aFile ? alacfile::parse_alac_m4a(J.file, again, c)
                                                                      : alacfile::parse_alac_caf(J.file, J.info, c);
            if (!err.empty()) {
                fprintf(stderr, " %s: \"%s\"\n", err.c_str(), J.in.c_str());
                return 1;
            }
            key = "D" + std::string(c.cookie.begin(), c.cookie.end());
        } else {
            char buf[64];
            snprintf(buf, sizeof(buf), "E%u/%u", J.info.bitsPerChannel, J.info.channels);
            key = buf;
        }
        groups[key].push_back(&J);
    }
    // one unit of work = the jobs of one group that one device takes
    struct Work {
        std::vector<Job *> jobs;
        std::vector<alacfile::AlacCafContents> contents;
        bool decode;
    };
    uint32_t workers = 1;
    int firstDevice = -1;  // -1: the classes' default (ALAC_HIP_DEVICE or 0), the single-device behaviour of every round before
    if (devices) {
        const int32_t have = alac_hip_device_count();
        // ALACCONVERT_SHARE_DEVICES=1 (tests on a one-GPU box): the N workers run side by side on the devices there are
        const bool share = getenv("ALACCONVERT_SHARE_DEVICES") != nullptr;
        if (have < 1 || ((int32_t)devices > have && !share)) {
            fprintf(stderr, " --devices %u: only %d GPU(s) visible\n", devices, have);
            return 1;
        }
        workers = devices;
        firstDevice = 0;
    }
    std::vector<std::vector<Work> > perWorker(workers);
    const int32_t visible = devices ? alac_hip_device_count() : 1;
    uint32_t next = 0;
    for (std::map<std::string, std::vector<Job *> >::iterator g = groups.begin(); g != groups.end(); ++g) {
        std::vector<Job *> &v = g->second;
        const bool dec = v[0]->info.isAlac;
        std::vector<Work> parts(workers);
        for (size_t j = 0; j < v.size(); j++) {
            Work &w = parts[(next + j) % workers];  // files dealt round-robin, continuing where the last group stopped
            w.decode = dec;
            w.jobs.push_back(v[j]);
            if (dec) {
                w.contents.push_back(alacfile::AlacCafContents());
                if (v[j]->info.kind == alacfile::kM4aFile) {
                    InputInfo again;
                    alacfile::parse_alac_m4a(v[j]->file, again, w.contents.back());
                } else {
                    alacfile::parse_alac_caf(v[j]->file, v[j]->info, w.contents.back());
                }
            }
        }
        for (uint32_t k = 0; k < workers; k++)
            if (!parts[(next + k) % workers].jobs.empty()) perWorker[(next + k) % workers].push_back(parts[(next + k) % workers]);
        next = (uint32_t)((next + v.size()) % workers);
    }
    std::vector<int> ok(workers, 1);
    auto run = [&](uint32_t k) {
        const int device = firstDevice < 0 ? -1 : (int)(k % (uint32_t)visible);
        for (size_t i = 0; i < perWorker[k].size() && ok[k]; i++) {
            Work &w = perWorker[k][i];
            ok[k] = w.decode ? decode_group(w.jobs, w.contents, device) : encode_group(w.jobs, segmentPackets, device);
        }
    };
    if (workers == 1) {
        run(0);
    } else {
        // one host thread and one context per device; nothing is shared between them (every Job belongs to one Work)
        std::vector<std::thread> threads;
        for (uint32_t k = 0; k < workers; k++) threads.emplace_back(run, k);
        for (size_t k = 0; k < threads.size(); k++) threads[k].join();
    }
    for (uint32_t k = 0; k < workers; k++)
        if (!ok[k]) return 1;
    for (size_t j = 0; j < jobs.size(); j++) {
        if (!alacfile::write_file(jobs[j].out, jobs[j].result)) {
            fprintf(stderr, " Cannot open file \"%s\"\n", jobs[j].out.c_str());
            return 1;
        }
    }
    return 0;
}
