#!/usr/bin/env python3
"""CPU port (oracle/alac_oracle.c) on every host core at once: one worker PROCESS per core, each encoding its own
slice of the synthetic workload as independent one-packet segments.  Never touches the GPU; bench.py runs it as
a child process and reads the one JSON line it prints (SURVEY.md §8d "all-cores run, core count printed").

    python tools/cpu_all_cores.py PACKETS BIT_DEPTH [CORES]
"""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def work(job):
    first, count, depth = job
    from alac_amd.capi import make_format, synth_pcm
    from oracle_lib import Oracle
    fmt = make_format(4096, depth, 2, 44100)
    pcm = synth_pcm(first, count, fmt)
    enc = Oracle().encoder(fmt.frame_size, fmt.bit_depth, fmt.num_channels, fmt.sample_rate)
    t0 = time.perf_counter()
    enc.encode_stream(pcm, count * fmt.frame_size, segment_packets=1)
    return time.perf_counter() - t0


def main():
    packets = int(sys.argv[1])
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    cores = int(sys.argv[3]) if len(sys.argv) > 3 else max(1, min(os.cpu_count() or 1, 64))
    per = (packets + cores - 1) // cores
    jobs = [(i * per, min(per, packets - i * per), depth) for i in range(cores) if i * per < packets]
    with mp.Pool(len(jobs)) as pool:
        pool.map(work, [(0, 1, depth)] * len(jobs))  # start the workers, load the libraries
        t0 = time.perf_counter()
        busy = pool.map(work, jobs)
        dt = time.perf_counter() - t0
    done = sum(j[1] for j in jobs)
    print(json.dumps(dict(value=done * 4096 / dt / 1e6, unit="Msamples/s", cores=len(jobs), kind="port",
                          sample=f"{done} packets over {len(jobs)} processes, wall {dt:.2f} s, "
                                 f"slowest worker {max(busy):.2f} s")))


if __name__ == "__main__":
    main()
