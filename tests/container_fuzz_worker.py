"""Worker of tests/test_container_fuzz.py: runs under LD_PRELOAD=libasan with convert-utility/libcontainer_asan.so — every
parser of the container code (sniff_input, parse_alac_caf, parse_alac_m4a, the sample-description and cookie un-wrappers) on
thousands of mutated and truncated files.  Any out-of-bounds read, overflow or other undefined behaviour aborts the process."""
import ctypes as C
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import container_lib as cl  # noqa: E402

cl.SO = os.path.join(cl.CU, "libcontainer_asan.so")


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    ct = cl.Container()
    rng = np.random.default_rng(11)
    cookie = bytes([0, 0, 16, 0, 0, 16, 40, 10, 14, 2, 0, 255]) + bytes(8) + struct.pack(">I", 44100)
    sizes = np.array([40, 33, 900, 7, 512], np.uint32)
    stream = bytes(rng.integers(0, 256, int(sizes.sum()), dtype=np.uint8))
    seeds = [ct.build_alac_m4a(44100, 2, 16, 4096 * 4 + 100, cookie, sizes, stream),
             ct.build_alac_caf(44100.0, 2, 16, 4096 * 4 * 4 + 400, cookie, sizes, stream),
             ct.build_wave(44100.0, 2, 16, stream), ct.build_wave(48000.0, 2, 24, stream, caf=True),
             ct.build_stsd(cookie, 2, 16, 44100), ct.cookie(0, cookie)]
    n = 0
    for r in range(rounds):
        base = bytearray(seeds[r % len(seeds)])
        kind = r % 5
        if kind == 0 and len(base) > 8:                       # truncation
            base = base[:int(rng.integers(0, len(base)))]
        elif kind == 1:                                       # random byte flips
            for _ in range(int(rng.integers(1, 12))):
                base[int(rng.integers(0, len(base)))] = int(rng.integers(0, 256))
        elif kind == 2:                                       # a 32-bit field set to a hostile value
            if len(base) >= 8:
                at = int(rng.integers(0, len(base) - 4))
                base[at:at + 4] = struct.pack(">I", int(rng.choice([0, 1, 7, 8, 0xffffffff, 0x7fffffff, 0x80000000, len(base), len(base) + 1])))
        elif kind == 3:                                       # splice two files
            other = seeds[int(rng.integers(0, len(seeds)))]
            cut = int(rng.integers(0, len(base) + 1))
            base = base[:cut] + bytearray(other[int(rng.integers(0, len(other))):])
        else:                                                 # garbage with a plausible head
            base = bytearray(base[:12]) + bytearray(rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8).tobytes())
        data = bytes(base)
        ct.sniff(data)
        ct.parse_alac_caf(data, max_packets=4096)
        ct.parse_alac_m4a(data, max_packets=4096)
        ct.parse_stsd(data)
        ct.cookie(1, data[:120])
        n += 1
    print("OK", n)
    return 0


if __name__ == "__main__":
    sys.exit(main())
