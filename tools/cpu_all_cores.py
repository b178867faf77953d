#!/usr/bin/env python3
"""CPU port (oracle/alac_oracle.c) on every USABLE host core at once: one worker PROCESS per core, each encoding its own
64-packet slice of the synthetic workload (all 8 signal classes, independent one-packet segments) over and over for
SECONDS of wall time.  Never touches the GPU; bench.py runs it as a child process and reads the one JSON line it prints
(SURVEY.md §8d "all-cores run, core count printed").

    python tools/cpu_all_cores.py [SECONDS=3] [BIT_DEPTH=16] [CORES]

Usable cores = min(affinity mask of this process, cgroup CPU quota, os.cpu_count()), capped at 64 workers: os.cpu_count()
alone reports the host's cores, not what the container may run on ("64 cores" with a 5.9 x speed-up).
"""
import json
import math
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SLICE = 64  # packets per worker slice: 8 of each signal class


def usable_cores():
    """-> (cores, how) from the affinity mask and the cgroup CPU quota (v2 cpu.max, v1 cfs_quota / cfs_period)"""
    host = os.cpu_count() or 1
    how = [f"os.cpu_count {host}"]
    n = host
    try:
        aff = len(os.sched_getaffinity(0))
        how.append(f"affinity {aff}")
        n = min(n, aff)
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            if q != "max":
                quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, p = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        how.append(f"cgroup quota {quota:.2f}")
        n = min(n, max(1, math.floor(quota)))
    return max(1, n), ", ".join(how)


def work(job):
    first, depth, seconds = job
    from alac_amd.capi import make_format, synth_pcm
    from oracle_lib import Oracle
    fmt = make_format(4096, depth, 2, 44100)
    pcm = synth_pcm(first, SLICE, fmt)
    enc = Oracle().encoder(fmt.frame_size, fmt.bit_depth, fmt.num_channels, fmt.sample_rate)
    done = 0
    t0 = time.perf_counter()
    while True:
        enc.encode_stream(pcm, SLICE * fmt.frame_size, segment_packets=1)
        done += SLICE
        dt = time.perf_counter() - t0
        if dt >= seconds:
            return done, dt


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    usable, how = usable_cores()
    cores = int(sys.argv[3]) if len(sys.argv) > 3 else min(usable, 64)
    with mp.Pool(cores) as pool:
        pool.map(work, [(0, depth, 0.0)] * cores)  # start the workers, load the libraries
        t0 = time.perf_counter()
        res = pool.map(work, [(i * SLICE, depth, seconds) for i in range(cores)], chunksize=1)
        wall = time.perf_counter() - t0
    done = sum(r[0] for r in res)
    one = max(r[0] / r[1] for r in res)  # fastest single worker, packets/s
    print(json.dumps(dict(value=done * 4096 / wall / 1e6, unit="Msamples/s", cores=cores, cores_usable=usable,
                          cores_how=how, kind="port",
                          speedup_vs_fastest_worker=round(done / wall / one, 2),
                          sample=f"{cores} worker processes, each looping over its own {SLICE}-packet slice for {seconds:.0f} s "
                                 f"({done} packets in {wall:.2f} s wall; slowest worker {max(r[1] for r in res):.2f} s)")))


if __name__ == "__main__":
    main()
