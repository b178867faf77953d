#!/usr/bin/env python3
"""bench.py — headline benchmark: batch ALAC encode of synthetic 44.1 kHz / 16-bit stereo packets.

    python bench.py --gpus N --steps K --warmup W

A step = R back-to-back passes (`--repeats`, default 16, stated in the line as config.encodes_per_step) of the encode hot
path (search + final predictor/entropy kernels + size scan + packer) over one batch of synthetic packets already resident
in HBM; R > 1 only lengthens the timed region (one pass at 10 000 packets is 1.5 ms: 20 of them would be a 30 ms sample).
  N = 1  BASELINE.json configs[1]: 10 000 independent 4096-sample 16-bit stereo packets; every packet of the timed
         workload is compared with the CPU oracle in the same run (bit_exact_vs_cpu) and the oracle is timed beside it.
         Legs beside the headline, each a few timed passes with its own roofline: `decode` (configs[4]), `bits24_10k`
         (configs[2]) and `shard_125k` (the per-GPU shard of configs[3], sampled oracle check + decode round trip).
  N > 1  BASELINE.json configs[3]: the 1 M-frame stream, 125 000 packets per rank (rank r = frames r*125000 ...), PCM
         generated ON THE DEVICE (alac_hip_synth_pcm), weak scaling; the shard bitstreams are re-assembled on every
         rank with RCCL (all-gather of shard sizes and packet sizes, then one grouped send/receive that places every
         shard at its prefix-sum offset) under the next pass's encode, timed by events on the side stream
         (`reassembly_ms`); every rank checks every 997th packet and its shard edges against the CPU oracle, rank 0
         checks the placement of every shard in the re-assembled stream.
         Started WITHOUT a torch.distributed environment (`python bench.py --gpus N`) this script launches its own N rank
         processes (`python -m torch.distributed.run ... bench.py ...` as a CHILD process, decided before torch or the
         GPU is touched) and relays rank 0's line; under `torch.distributed.run` it is a rank.
`--packets P` overrides the per-rank batch (e.g. `--gpus 1 --packets 125000` runs the configs[3] shard shape on one GPU).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMDS = 256 * 4                      # MI355X: 256 CUs x 4 SIMDs
ISSUE_PEAK_GINST = SIMDS * 2.4 / 4   # wave-instructions/ns the chip can issue: one per SIMD per 4 cycles at 2.4 GHz
                                     # (tools/op_rate_microbench.hip: every integer op of these kernels except plain
                                     # add / sub / and / xor / mov occupies its SIMD for ~4.3 cycles per wave64)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=16,
                    help="encode passes per step (the timed region is steps x repeats passes; stated in the JSON line)")
    ap.add_argument("--packets", type=int, default=0,
                    help="packets per GPU per pass (default: 10000 = configs[1] at N = 1, 125000 = configs[3] at N > 1)")
    ap.add_argument("--bit-depth", type=int, default=16)
    ap.add_argument("--cpu-packets", type=int, default=10000, help="size of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-decode", action="store_true", help="skip the decode-direction measurement (N = 1)")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the bits24_10k / shard_125k legs (N = 1); profiling runs use this so that one command = one shape")
    ap.add_argument("--no-reassemble", action="store_true", help="skip the RCCL re-assembly at N > 1")
    ap.add_argument("--force-reassemble", action="store_true",
                    help="rehearsal: run the RCCL re-assembly path even with one rank (needs torch.distributed.run)")
    return ap.parse_args(argv)


# ---- self-launch (N > 1 from the plain command) ------------------------------------------------------------------------
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(n, argv, port):
    """what `python bench.py --gpus N ...` turns into: one rank per GPU of this node, rendezvous on 127.0.0.1"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv, run=subprocess.run):
    """Start the N rank processes as a child (never a re-exec: this process has not imported torch and must not touch the
    GPU), relay rank 0's JSON line as this process's single stdout line, everything else to stderr.  Returns the exit code:
    the child's if non-zero, 3 if the ranks printed no result line."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    p = run(launch_command(args.gpus, argv, free_port()), stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    line = None
    for ln in (p.stdout or "").splitlines():
        s = ln.strip()
        if s.startswith("{") and '"metric"' in s:
            try:
                json.loads(s)
                line = s
                continue
            except ValueError:
                pass
        if s:
            print(ln, file=sys.stderr)
    if p.returncode != 0:
        print(f"bench.py: the rank processes exited with code {p.returncode}", file=sys.stderr)
        return p.returncode
    if line is None:
        print("bench.py: the rank processes printed no result line", file=sys.stderr)
        return 3
    print(line, flush=True)
    return 0


def stub_rank_main(args):
    """ALAC_BENCH_STUB=1 (tests/test_bench_launcher.py, no GPU): the rank side of the launcher contract on gloo — rendezvous,
    barrier, max over ranks, rank 0 prints one line.  ALAC_BENCH_STUB=fail: rank 1 exits non-zero."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if os.environ.get("ALAC_BENCH_STUB") == "fail" and rank == world - 1:
        os._exit(7)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    print(f"rank {rank} chatter that is not the result line")
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": float(t.item()), "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "stub": True}), flush=True)
    dist.destroy_process_group()
    return 0


# ---- CPU legs -----------------------------------------------------------------------------------------------------------
def cpu_baseline(fmt, pcm_host, packets, gpu_stream=None, gpu_sizes=None, min_seconds=10.0, max_passes=8):
    """Time the CPU oracle (single thread, 'port') on a bounded sample of the same workload — whole passes over the
    first `packets` packets until at least `min_seconds` of CPU work — and, as a by-product, check the GPU bytes of
    that sample against it.  Where the reference's own compiled stage objects travelled with the repo (oracle/_ref),
    one more pass drives THEM (pc_block, dyn_comp of codec/dp_enc.c / ag_enc.c) under the same restated packet driver."""
    import numpy as np
    from oracle_lib import Oracle, Ref, have_ref
    o = Oracle()
    nbytes = packets * fmt.packet_bytes

    def one_pass(hooks=None):
        enc = o.encoder(fmt.frame_size, fmt.bit_depth, fmt.num_channels, fmt.sample_rate, hooks=hooks)
        t0 = time.perf_counter()
        r, z = enc.encode_stream(pcm_host[:nbytes], packets * fmt.frame_size, segment_packets=1)
        return time.perf_counter() - t0, r, z

    times = []
    ref = ref_sizes = None
    while len(times) < max_passes and (not times or sum(times) < min_seconds):
        dt, ref, ref_sizes = one_pass()
        times.append(dt)
    mean = sum(times) / len(times)
    exact = None
    if gpu_stream is not None:
        n = int(ref_sizes.astype(np.int64).sum())
        exact = bool(np.array_equal(gpu_sizes[:packets], ref_sizes) and np.array_equal(gpu_stream[:n], ref))
    base = dict(value=packets * fmt.frame_size / mean / 1e6, unit="Msamples/s", cores=1, kind="port",
                sample=f"{len(times)} passes over the first {packets} packets of the workload ({sum(times):.1f} s of CPU work, "
                       f"fastest pass {min(times):.2f} s), codec stages only (PCM in RAM -> packets in RAM), "
                       f"oracle/alac_oracle.c -O2 single thread")
    stages = None
    if have_ref():
        try:
            H = Ref().hooks()
            dt, r2, z2 = one_pass(H)
            stages = dict(value=packets * fmt.frame_size / dt / 1e6, unit="Msamples/s", cores=1, kind="reference stage objects",
                          same_bytes=bool(np.array_equal(r2, ref) and np.array_equal(z2, ref_sizes)),
                          sample=f"one pass over the same {packets} packets with the reference's own compiled pc_block / "
                                 f"dyn_comp (oracle/_ref, -O2) under the restated packet driver, {dt:.2f} s")
        except Exception as e:  # reported, never fatal for the headline
            stages = {"error": repr(e)}
    return base, exact, stages


def cpu_all_cores(fmt, seconds=3.0):
    """The same CPU port on every USABLE host core at once (tools/cpu_all_cores.py: one worker process per core of the
    affinity mask / cgroup quota, each looping over its own slice for `seconds`; run as a child process so that nothing
    here forks after the GPU is initialised).  SURVEY.md §8d."""
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_all_cores.py"), str(seconds), str(fmt.bit_depth)],
                           capture_output=True, text=True, timeout=300)
        return json.loads(p.stdout.strip().splitlines()[-1])
    except Exception as e:  # the headline does not depend on it
        return {"error": repr(e)}


def baseline_metric():
    """The metric string of BASELINE.json (falls back to its text if the file is not there)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "encode Msamples/s (bit-exact) 44.1kHz/16-bit stereo, 1/2/4/8 GPU"


def kernel_symbol(stage, fused, depth, thru=False):
    """HIP kernel behind a bench stage name (what rocprofv3's kernel stats list)."""
    if thru:  # throughput regime: every stage fills the machine, one lane per chain, predictor and Golomb work of a chain in ONE lane
        return {"lms_search1": f"k_search1_lane<{depth}> (five mixRes pc_block passes + their dyn_comp counts in the lane)",
                "golomb_count1": "(inside k_search1_lane)",
                "lms_search2": f"k_search2_lane<{depth}, 2> (converge passes + numUV counts in the lane)",
                "golomb_count2": "(inside k_search2_lane)",
                "lms_final": f"k_class_final<{depth}, 2, 8> || k_class_final<{depth}, 2, 4> (final pc_block pass + final dyn_comp of a chain in ONE "
                             "lane, per packet class, side by side on two streams; + k_decide2, k_class_count, k_class_assign)",
                "golomb_final": "(inside k_class_final)",
                "finalize_scan": "k_finalize + k_scan_sizes", "pack": "k_pack"}.get(stage, stage)
    # latency regime: producer / consumer launches of 4-wave worker workgroups
    return {"lms_search1": f"k_search1_fused<{depth}, 4, 2, true> (mixRes search passes || their dyn_comp counts, one launch of workers)",
            "golomb_count1": "(inside k_search1_fused)",
            "lms_search2": f"k_lms_search2_w<{depth}, 2>", "golomb_count2": "k_gol_count2_w<2>",
            "lms_final": f"k_final_fused<{depth}, 2, 4, 2, false, true> (final pc_block pass || final dyn_comp, numUV / escape decision and "
                         "packet sizes folded in, one launch of workers)",
            "golomb_final": "(inside k_final_fused)",
            "finalize_scan": "k_scan_sizes", "pack": "k_pack"}.get(stage, stage)


def load_counters(name, key):
    """A table of profiles/<name> — only if it was collected on THESE kernel sources (the collecting script stores
    alac_amd.source_fingerprint() beside every table; a kernel edit without a profile refresh must not leave stale counters
    in the line).  -> (table or None, note)"""
    import alac_amd
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            doc = json.load(f)
    except Exception:
        return None, f"profiles/{name} absent"
    if key not in doc:
        return None, f"no table {key} in profiles/{name}"
    want, have = alac_amd.source_fingerprint(), doc.get("_fingerprint", {}).get(key)
    if have != want:
        return None, (f"stale: profiles/{name}[{key}] was collected on kernel sources {have}, this run is {want} "
                      "(refresh with tools/profile_round.sh)")
    return doc[key], f"profiles/{name}[{key}] (rocprofv3 --pmc, same command, kernel sources {want})"


def sampled_oracle_check(alac_amd, fmt, first_frame, B, stream_dev, offsets_dev, sizes_dev, every=997):
    """about every 1000th packet of the shard plus its edges, GPU bytes vs the CPU oracle on the
    same (host-generated) frames.  The stride is 997, not 1000: the generator's 8 signal classes go by frame index mod 8,
    and a stride that is a multiple of 8 would only ever sample silence.  Returns (packets checked, all equal)."""
    import numpy as np
    from oracle_lib import Oracle
    idx = sorted(set(list(range(0, B, every)) + [0, 1, B // 2, B - 2, B - 1]) & set(range(B)))
    offs = offsets_dev.cpu().numpy()
    sizes = sizes_dev.cpu().numpy()
    enc = Oracle().encoder(fmt.frame_size, fmt.bit_depth, fmt.num_channels, fmt.sample_rate)
    ok = True
    for p in idx:
        pcm = alac_amd.synth_pcm(first_frame + p, 1, fmt)
        enc.reset()
        want = enc.encode_packet(pcm, fmt.frame_size)
        got = stream_dev[int(offs[p]):int(offs[p + 1])].cpu().numpy()
        ok = ok and int(sizes[p]) == len(want) and np.array_equal(got, want)
    return len(idx), bool(ok)


def shard_checksum(torch, x):
    """Placement check of the re-assembled stream, computed on the device: (sum of bytes, position-weighted sum) of a
    shard, with positions relative to the shard's own start."""
    v = x.to(torch.int64)
    w = (torch.arange(v.numel(), device=x.device, dtype=torch.int64) % 65521) + 1
    return torch.stack([v.sum(), (v * w).sum()])


def encode_roofline(fmt, B, total_bytes, stage_ms, thru_hint):
    """stages / roofline of one encode shape from the library's own HIP-event stage timing (alac_hip_profile_*: events
    recorded on the context's stream around every stage of the timed passes).
    Algorithmic bytes per launch, SURVEY.md §8(d).  A FUSED launch (predictor + entropy coder in one kernel, or both in one
    lane) is charged the compulsory bytes only: PCM in + packet bits out; the residual planes its two halves hand to each
    other through HBM are reported separately as hand-off bytes (waste, not work).  Stand-alone stages keep the §8(d)
    stand-alone figures (LPC+mix: PCM in + int32 residuals out; entropy: residuals in + bits out)."""
    depth = fmt.bit_depth
    n8 = fmt.frame_size // 8
    bps = 3 if depth in (20, 24) else depth // 8
    res_full = B * 2 * fmt.frame_size * 4
    res_s1 = B * 2 * 5 * n8 * 4
    # what a stage moves algorithmically (its own inputs + outputs), for the per-stage GB/s column ...
    moved = {
        "lms_search1": B * 2 * n8 * bps + res_s1,
        "golomb_count1": res_s1,
        "lms_search2": B * 2 * (n8 // 4) * bps + B * 4 * (n8 // 4) * 4,
        "golomb_count2": B * 4 * n8 * 4,
        "lms_final": B * fmt.packet_bytes + res_full,
        "golomb_final": res_full + total_bytes,
        "finalize_scan": B * (64 + 4 + 8),
        "pack": 2 * total_bytes,
    }
    # ... and the COMPULSORY bytes the roofline is priced on, SURVEY §8(d): "search passes add zero compulsory traffic
    # (data already resident)"
    algo = dict(moved)
    for k in ("lms_search1", "golomb_count1", "lms_search2", "golomb_count2"):
        algo[k] = 0
    handoff = {k: 0 for k in algo}
    # Launches that run a predictor stage AND its entropy stage report their time under the lms_* stage (the golomb_* stage
    # is then an empty event interval).
    fused = []
    for a, b, compulsory, ho in (("lms_search1", "golomb_count1", 0, 2 * res_s1),
                                 ("lms_search2", "golomb_count2", 0, 2 * B * 4 * (n8 // 4) * 4),
                                 ("lms_final", "golomb_final", B * fmt.packet_bytes + total_bytes, 2 * res_full)):
        if stage_ms[a][0] > 0 and stage_ms[b][0] < 0.05 * stage_ms[a][0]:
            algo[a], algo[b] = compulsory, 0
            moved[a], moved[b] = moved[a] + moved[b] - (ho // 2), 0
            handoff[a] = ho
            fused.append(a + "+" + b)
    thru = bool(thru_hint)
    if thru:
        # one lane per chain does the predictor AND the Golomb work of its chain: the residuals never leave the CU, except the
        # mixRes = 4 search pass (one of five planes), which the numUV decision reads behind the converge passes
        for k in fused:
            handoff[k.split("+")[0]] = 0
        if "lms_search1+golomb_count1" in fused:
            handoff["lms_search1"] = 2 * (res_s1 // 5)
    # stage_ms[k] = (mean ms of one launch, launches per pass).  The dominant kernel among the stages that HAVE compulsory
    # traffic (a search kernel can be the longest single launch; its §8(d) bytes are zero: no HBM roofline to carry)
    dom = max((k for k in stage_ms if algo[k] > 0), key=lambda k: stage_ms[k][0] * max(stage_ms[k][1], 1))
    longest = max(stage_ms, key=lambda k: stage_ms[k][0] * max(stage_ms[k][1], 1))
    ms_dom, n_dom = stage_ms[dom][0], max(stage_ms[dom][1], 1)
    algo_bytes = algo[dom] // n_dom
    achieved = algo_bytes / (ms_dom * 1e-3) / 1e9 if ms_dom > 0 else 0.0
    stages = {k: {"ms_per_launch": round(v[0], 4), "launches_per_step": v[1],
                  "moved_GBps": round(moved[k] / max(v[1], 1) / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 else None}
              for k, v in stage_ms.items()}
    gpu_ms = sum(v[0] * max(v[1], 1) for v in stage_ms.values())
    compulsory_step = B * fmt.packet_bytes + total_bytes
    key = f"{depth}bit_stereo_{B}"
    ttab, tnote = load_counters("hbm_traffic.json", key)
    traffic = ttab.get(dom) if ttab else None
    traffic_step = sum(v for v in ttab.values() if isinstance(v, (int, float))) if ttab else None
    upper = (ttab or {}).get("_upper_bound") or {}
    traffic_upper = upper.get(dom) if upper else None
    traffic_step_upper = sum(upper.values()) if upper else None
    # the bound that actually holds: VALU issue.  Wave-instructions per pass (PMC instruction mix of the same command)
    issue = None
    itab, inote = load_counters("instruction_mix.json", key)
    if itab and gpu_ms > 0:
        tot = itab.get("wave_instructions_per_step")
        if tot:
            ach = tot / (gpu_ms * 1e-3) / 1e9
            issue = {"bound": "valu-issue", "scope": "whole pass (all kernels)", "achieved": round(ach, 1),
                     "peak": round(ISSUE_PEAK_GINST, 1), "unit": "G wave-instructions/s", "frac": round(ach / ISSUE_PEAK_GINST, 3),
                     "wave_instructions_per_step": int(tot), "source": inote}
            per = itab.get("per_stage", {}).get(dom)
            if per and ms_dom > 0:
                a2 = per / n_dom / (ms_dom * 1e-3) / 1e9
                issue["dominant_kernel"] = {"achieved": round(a2, 1), "frac": round(a2 / ISSUE_PEAK_GINST, 3),
                                            "wave_instructions_per_launch": int(per / n_dom)}
    else:
        issue = {"source": inote}
    sym = kernel_symbol(dom, fused, depth, thru)
    roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": tnote,
            "traffic_whole_step": traffic_step,
            "traffic_upper_bound": traffic_upper, "traffic_whole_step_upper_bound": traffic_step_upper,
            "traffic_how": "reads = FETCH_SIZE x the factor calibrated on the kernel's read shape (2.0 coalesced streams and the "
                           "16-bit predictor staging, 1.707 the 20- / 24-bit staging, 1.43 the entropy decoder's word stream; tools/fetch_calibrate.hip as "
                           "MI355X_MICROARCH.md asks for access shapes other than wide streaming reads), writes = WRITE_SIZE; *_upper_bound = 2 x FETCH_SIZE "
                           "+ WRITE_SIZE whatever the shape (what rounds 1-3 reported)",
            "kernel": dom, "kernel_symbol": sym, "longest_stage": longest,
            "kernel_ms": round(ms_dom, 4), "launches_per_step": n_dom,
            "algorithmic_bytes_per_launch": algo_bytes,
            "handoff_bytes": handoff[dom] // n_dom,
            "whole_step": {"algorithmic_bytes": compulsory_step, "gpu_ms": round(gpu_ms, 4),
                           "achieved": round(compulsory_step / (gpu_ms * 1e-3) / 1e9, 2) if gpu_ms > 0 else None,
                           "frac": round(compulsory_step / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if gpu_ms > 0 else None,
                           "traffic_over_algorithmic": round(traffic_step / compulsory_step, 2) if traffic_step else None,
                           "traffic_upper_bound_over_algorithmic": round(traffic_step_upper / compulsory_step, 2) if traffic_step_upper else None},
            "issue": issue,
            "note": "dominant stage by measured time (HIP events on the library's stream inside this run); algorithmic bytes = "
                    "SURVEY §8(d) compulsory bytes (PCM in + packet bits out for a fused launch), hand-off planes counted "
                    "separately; `traffic` = PMC bytes per launch of that stage summed per pass; the path is bound by VALU issue on "
                    "serial integer recurrences (sign-LMS, Golomb mean tracker), not by HBM"}
    return stages, fused, roof


def decode_leg(torch, ctx, fmt, B, stream, offsets, d_pcm, reps, warm=1):
    """decode direction (BASELINE configs[4]): the packed stream back to PCM, round trip checked against the generator output;
    roofline on the compulsory bytes of the whole decode pass (stream in + PCM out), timed with HIP events on the context's
    stream"""
    cookie = ctx.magic_cookie(fmt)
    # outputs allocated once: a loop that allocates 2 GB per call times the caching allocator, not the decoder
    d_bufs = (torch.empty(B * fmt.packet_bytes, dtype=torch.uint8, device="cuda"),
              torch.zeros(B, dtype=torch.int32, device="cuda"), torch.zeros(B, dtype=torch.int32, device="cuda"))
    with torch.cuda.stream(ctx.stream):
        for _ in range(max(1, warm)):  # untimed: the leg may follow seconds of host-side checking (clocks down, cold caches)
            ctx.decode(cookie, stream, offsets, B, out=d_bufs)
        ctx.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t1 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            d_out, d_ns, d_st, _ = ctx.decode(cookie, stream, offsets, B, out=d_bufs)  # every packet is a full frame
        e1.record()
        ctx.synchronize()
    ddt = (time.perf_counter() - t1) / reps
    ev_ms = e0.elapsed_time(e1) / reps
    total = int(offsets[-1].item())
    comp = total + B * fmt.packet_bytes
    ach = comp / (ev_ms * 1e-3) / 1e9
    ttab, tnote = load_counters("hbm_traffic.json", f"decode_{fmt.bit_depth}bit_stereo_{B}")
    tstep = sum(v for v in ttab.values() if isinstance(v, (int, float))) if ttab else None
    tupper = sum(((ttab or {}).get("_upper_bound") or {}).values()) or None
    return {"ms_per_step": round(ddt * 1e3, 4), "value": round(B * fmt.frame_size / ddt / 1e6, 1),
            "unit": "Msamples/s", "steps": reps, "warmup": max(1, warm),
            "round_trip_exact": bool(torch.equal(d_out, d_pcm)) and int(d_st.abs().sum()) == 0,
            "roofline": {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": tstep, "traffic_upper_bound": tupper,
                         "traffic_source": tnote,
                         "kernel": "whole decode pass (k_dec_stage, k_dec_header, then up to 65 536 chains k_dec_fused_wg [entropy wave + "
                                   "its three predictor waves per workgroup], above that k_dec_raw, k_dec_entropy_wide, k_dec_unpc_wide "
                                   "[pairs write the PCM]; k_dec_unpc, k_dec_unmix for what is left)",
                         "kernel_ms": round(ev_ms, 4), "algorithmic_bytes_per_launch": comp,
                         "note": "compulsory = packed stream in + PCM out; HIP events on the context's stream around the timed "
                                 "passes; bound by the serial adaptive-Golomb bit walk per packet, not by HBM"}}


def encode_leg(torch, alac_amd, ctx, fmt, B, passes=6, warm=2, every=997, with_decode=True):
    """a few timed passes of another shape beside the headline (same library calls, same stream discipline)"""
    d_pcm = ctx.synth_pcm(0, B, fmt)
    bufs = ctx.encode_buffers(fmt, B)
    with torch.cuda.stream(ctx.stream):
        for _ in range(warm):
            ctx.encode(fmt, d_pcm, B, bufs=bufs)
        ctx.synchronize()
        ctx.profile_begin(min(passes, 3))
        t0 = time.perf_counter()
        for _ in range(passes):
            ctx.encode(fmt, d_pcm, B, bufs=bufs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / passes
        calls, stage_ms = ctx.profile_end()
        ctx.synchronize()
    total = int(bufs["offsets"][-1].item())
    thru = ctx.regime(fmt, B) == "throughput"
    stages, fused, roof = encode_roofline(fmt, B, total, stage_ms, thru)
    n_chk, ok = sampled_oracle_check(alac_amd, fmt, 0, B, bufs["out"], bufs["offsets"], bufs["sizes"], every=every)
    out = {"workload": f"{B} independent 4096-sample {fmt.bit_depth}-bit stereo packets, generated on the device",
           "ms_per_step": round(dt * 1e3, 4), "value": round(B * fmt.frame_size / dt / 1e6, 1), "unit": "Msamples/s",
           "steps": passes, "warmup": warm, "regime": ctx.regime(fmt, B), "output_bytes": total,
           "bit_exact_sampled": {"packets_checked": n_chk, "equal": ok,
                                 "what": f"every {every}th packet (all 8 signal classes) + edges, GPU bytes vs CPU oracle"},
           "stages": stages, "fused_launches": fused, "roofline": roof}
    if with_decode:
        out["decode"] = decode_leg(torch, ctx, fmt, B, bufs["out"], bufs["offsets"], d_pcm, reps=10, warm=3)
    del d_pcm, bufs
    torch.cuda.empty_cache()
    return out


class CabiReassembler:
    """The re-assembly of bench.py's N > 1 loop on the C-ABI's communicator (alac_amd.Comm -> alac_hip_comm_*,
    alac_hip_reassemble_begin / _finish): torch.distributed only carries the 128-byte ncclUniqueId to the ranks."""

    def __init__(self, torch, dist, alac_amd, device, rank, world, packets, shard_capacity):
        idt = torch.zeros(alac_amd.capi.COMM_ID_BYTES, dtype=torch.uint8, device="cuda")
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(alac_amd.Comm.unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        torch.cuda.synchronize()
        self.comm = alac_amd.Comm(device, bytes(idt.cpu().numpy().tobytes()), rank, world)
        self.world, self.packets = world, packets
        self.stream = torch.empty(world * shard_capacity, dtype=torch.uint8, device="cuda")
        self.sizes = torch.empty(world * packets, dtype=torch.int32, device="cuda")
        self.slot = 0

    def begin(self, b):
        slot, self.slot = self.slot, (self.slot + 1) % alac_amd_slots()
        self.comm.begin(slot, b["offsets"], self.packets, b["out"].numel(), self.stream.numel(), sizes=b["sizes"],
                        all_sizes=self.sizes)
        return slot

    def finish(self, slot, b):
        offs = self.comm.finish(slot, b["out"], self.stream)
        return dict(stream=self.stream, total=offs[-1], offsets=offs, sizes=self.sizes,
                    lens=[offs[r + 1] - offs[r] for r in range(self.world)],
                    mode="C-ABI: ncclAllGather of the shard table + one ncclSend/ncclRecv group")


class TorchReassembler:
    """The same exchange on torch.distributed (alac_amd.reassemble.Reassembler) behind CabiReassembler's interface: only used
    when the C-ABI communicator could not be created on EVERY rank (a collective decision, see make_reassembler) — the line
    then says so in `reassembly_path`."""

    def __init__(self, dist):
        from alac_amd.reassemble import Reassembler
        self.r = Reassembler(dist.group.WORLD)

    def begin(self, b):
        return self.r.begin(b["out"], b["offsets"][-1:], b["sizes"])

    def finish(self, h, b):
        g = self.r.finish(h)
        offs = [int(x) for x in g["offsets"].tolist()]
        return dict(stream=g["stream"], total=g["total"], offsets=offs, sizes=g["sizes"],
                    lens=[offs[r + 1] - offs[r] for r in range(len(offs) - 1)],
                    mode="torch.distributed fallback: " + g["mode"])


def make_reassembler(torch, dist, alac_amd, device, rank, world, packets, shard_capacity):
    """C-ABI communicator on every rank, or — if ANY rank could not create it — torch.distributed on every rank.  Creation is
    itself collective (ncclCommInitRank), so a rank only gets to fail before it: a missing librccl, a refused argument; the
    verdict is all-reduced so that no rank runs a different exchange from its peers."""
    ra, err = None, ""
    try:
        if os.environ.get("ALAC_BENCH_NO_CABI_COMM"):  # rehearsal of the fallback (set on every rank or on one)
            raise RuntimeError("ALAC_BENCH_NO_CABI_COMM is set")
        ra = CabiReassembler(torch, dist, alac_amd, device, rank, world, packets, shard_capacity)
    except Exception as e:  # noqa: BLE001 - reported in the line
        err = f"{type(e).__name__}: {e}"
    ok = torch.tensor([1 if ra is not None else 0], dtype=torch.int32, device="cuda")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 1:
        return ra, "C-ABI (alac_hip_comm_* / alac_hip_reassemble_begin/_finish on librccl)"
    print(f"bench.py rank {rank}: C-ABI communicator unavailable on some rank ({err or 'another rank'}); "
          "re-assembly falls back to torch.distributed", file=sys.stderr)
    return TorchReassembler(dist), "torch.distributed fallback (" + (err or "another rank failed") + ")"


def alac_amd_slots():
    from alac_amd.capi import COMM_SLOTS
    return COMM_SLOTS


def rank_main(args):
    # stdout carries ONE JSON line and nothing else: native libraries write banners to fd 1 (RCCL prints its version block
    # at the first communicator), so fd 1 points at stderr for the duration of the run and the line goes to the saved fd
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        return rank_main_body(args, json_fd)
    finally:
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        os.close(json_fd)


def rank_main_body(args, json_fd):
    import numpy as np
    import torch
    import torch.distributed as dist

    import alac_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_reassemble
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    R = max(1, args.repeats)
    fmt = alac_amd.make_format(4096, args.bit_depth, 2, 44100)
    B = args.packets if args.packets > 0 else (10000 if world == 1 else 125000)
    first_frame, mine = alac_amd.shard_range(world * B, world, rank)  # weak scaling: B packets on every rank
    assert mine == B
    ctx = alac_amd.Context(local_rank)

    # synthetic input, generated ON THE DEVICE (alac_hip_synth_pcm: same source and bytes as the host generator the
    # oracle legs use, tests/test_gpu_synth.py) and resident in HBM before any timing
    d_pcm = ctx.synth_pcm(first_frame, B, fmt)
    ctx.synchronize()
    # N > 1: the re-assembly of pass i runs on a side stream under the encode of pass i+1.  It goes through the C-ABI
    # (alac_hip_reassemble_begin / _finish, alac_comm.cpp: librccl called from C++, no torch.distributed on the data
    # path) in two phases so that the host never waits for the GPU between two encodes: the shard lengths of pass i are
    # exchanged right after its encode, the shard bytes one host step later.  Three output buffer sets rotate because a
    # shard must stay untouched until its sends have run.
    reassemble = use_dist and not args.no_reassemble
    nbuf = 3 if reassemble else 2
    bufs = [ctx.encode_buffers(fmt, B) for _ in range(nbuf)]
    comm_stream = torch.cuda.Stream() if reassemble else None
    ra, ra_path = (make_reassembler(torch, dist, alac_amd, local_rank, rank, world, B, bufs[0]["out"].numel()) if reassemble
                   else (None, None))
    state = {"pending": None, "gather": None, "timing": False, "ev": [], "on": reassemble}

    def timed(fn):
        """run fn on the current (side) stream between two timing events (kept for reassembly_ms)"""
        if not state["timing"]:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        state["ev"].append((a, b))
        return r

    def finish_pending():
        if state["pending"] is None:
            return
        h, b = state["pending"]
        with torch.cuda.stream(comm_stream):
            state["gather"] = timed(lambda: ra.finish(h, b))
            done = torch.cuda.Event()
            done.record()
        b["done"] = done  # the buffer set may be encoded into again once this has run
        state["pending"] = None

    def one_pass(i):
        b = bufs[i % nbuf]
        if "done" in b:
            torch.cuda.current_stream().wait_event(b.pop("done"))
        ctx.encode(fmt, d_pcm, B, bufs=b)
        if state["on"]:
            ev = torch.cuda.Event()
            ev.record()
            finish_pending()  # bytes of pass i-1: runs under the encode of pass i just launched
            comm_stream.wait_event(ev)
            with torch.cuda.stream(comm_stream):
                state["pending"] = (timed(lambda: ra.begin(b)), b)
        return b

    # the passes are issued ON the context's stream: the events one_pass() records for the hand-over to the RCCL side
    # stream are then ordered behind the encode without any cross-stream wait between two encodes
    with torch.cuda.stream(ctx.stream):
        for i in range(args.warmup * R):
            one_pass(i)
        finish_pending()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()

    n_pass = args.steps * R
    # stage events (HIP events between the launches, on the library's stream) of the first 16 timed passes only: an event between
    # two launches costs 5-12 us of idle GPU (rocprofv3 kernel trace: back-to-back launches have no gap without them)
    ctx.profile_begin(min(n_pass, 16))
    state["timing"] = True
    t0 = time.perf_counter()
    with torch.cuda.stream(ctx.stream):
        for i in range(n_pass):
            last = one_pass(args.warmup * R + i)
        finish_pending()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    state["timing"] = False
    calls, stage_ms = ctx.profile_end()
    ctx.synchronize()  # raises if an in-launch hand-off of any timed pass was lost (the outputs would be invalid)

    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    reasm_ms = None
    if reassemble and state["ev"]:
        # per pass: the length exchange (begin) + the shard bytes (finish), device time on the side stream
        tot = sum(a.elapsed_time(b) for a, b in state["ev"])
        t = torch.tensor([tot / n_pass], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        reasm_ms = float(t.item())

    # N > 1: the same K steps again WITHOUT the exchange (the --no-reassemble figure), so that one line carries both the
    # overlapped value and the encode-only value and a scaling curve can be split into encode and RCCL time
    dt_plain = None
    if reassemble:
        state["on"] = False
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        with torch.cuda.stream(ctx.stream):
            for i in range(n_pass):
                one_pass(args.warmup * R + n_pass + i)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        tp = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
        dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        dt_plain = float(tp.item())
        ctx.synchronize()
        state["on"] = True

    total_bytes = int(last["offsets"][-1].item())

    # ---- verification legs shared by every rank (after the timed region) -----------------------------------------
    big = B > args.cpu_packets or world > 1 or args.cpu_packets == 0
    sampled = None
    if big:
        n_chk, ok = sampled_oracle_check(alac_amd, fmt, first_frame, B, last["out"], last["offsets"], last["sizes"])
        sampled = {"packets_checked_per_rank": n_chk, "equal": ok,
                   "what": "every 997th packet (all 8 signal classes) + shard edges, GPU bytes vs CPU oracle on host-generated frames"}
    placement = None
    if use_dist:
        ok_all = torch.tensor([1 if (sampled is None or sampled["equal"]) else 0], dtype=torch.int64, device="cuda")
        dist.all_reduce(ok_all, op=dist.ReduceOp.MIN)
        if sampled is not None:
            sampled["equal_on_every_rank"] = bool(ok_all.item())
        mine_sum = shard_checksum(torch, last["out"][:total_bytes])
        sums = torch.empty(world * 2, dtype=torch.int64, device="cuda")
        dist.all_gather_into_tensor(sums, mine_sum)
        gather = state["gather"]
        if reassemble and gather is not None:
            offs = gather["offsets"]
            placed = all(bool(torch.equal(shard_checksum(torch, gather["stream"][int(offs[r]):int(offs[r + 1])]), sums[2 * r:2 * r + 2]))
                         for r in range(world))
            pakt_ok = gather["sizes"] is not None and bool(torch.equal(gather["sizes"][rank * B:(rank + 1) * B], last["sizes"])) \
                and int(gather["sizes"].to(torch.int64).sum().item()) == gather["total"]
            placement = {"shards_at_prefix_sum_offsets": placed, "packet_size_table_consistent": pakt_ok,
                         "exchange": gather.get("mode"),
                         "stream_bytes": gather["total"], "shard_bytes": [int(x) for x in gather["lens"]]}

    if rank == 0:
        samples = world * B * fmt.frame_size * n_pass
        value = samples / dt / 1e6
        thru = ctx.regime(fmt, B) == "throughput"
        stages, fused, roof = encode_roofline(fmt, B, total_bytes, stage_ms, thru)
        cfg_name = "configs[1]" if (world == 1 and B == 10000) else ("configs[3] shard shape" if B == 125000 else "custom batch")
        out = {
            "metric": baseline_metric(),
            "value": round(value, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32" if args.bit_depth != 16 else "int32 (int16 PCM, int16 coefficients)",
            "data": "synthetic",
            "config": {"workload": f"{B} independent 4096-sample {args.bit_depth}-bit stereo packets per GPU per encode pass "
                                   f"(BASELINE.json {cfg_name}"
                                   + (f": frames {0}..{world * B - 1} of the 1 M-frame stream, rank r = frames r*{B}.." if world > 1 else "")
                                   + f"), 8 deterministic signal classes, generated on the device, resident in HBM; a step = "
                                     f"{R} back-to-back encode passes over it",
                       "packets_per_gpu": B, "frame_size": 4096, "bit_depth": args.bit_depth, "channels": 2,
                       "encodes_per_step": R,
                       "segments": "one packet per segment (state = init_coefs)",
                       "rccl_ranks": dist.get_world_size() if use_dist else 0,
                       "reassembly": ("none (1 GPU)" if not use_dist else
                                      ("skipped" if args.no_reassemble else "RCCL behind the C-ABI (alac_hip_reassemble_begin/_finish): "
                                       "all-gather of shard lengths + packet sizes, one group of ncclSend/ncclRecv of the shard "
                                       "bytes straight to their prefix-sum offsets, overlapped with the next pass's encode"))},
            "ms_per_encode": round(dt / n_pass * 1e3, 4),
            "packets_per_s": round(world * B * n_pass / dt, 1),
            "x_realtime": round(value * 1e6 / 44100.0, 1),
            "output_bytes_per_encode_per_gpu": total_bytes,
            "stages": stages,
            "fused_launches": fused,
            "regime": ctx.regime(fmt, B),
            "calls_timed": n_pass,
            "calls_with_stage_events": calls,
            "roofline": roof,
        }
        if dt_plain is not None:
            out["value_no_reassemble"] = round(samples / dt_plain / 1e6, 3)
            out["ms_per_encode_no_reassemble"] = round(dt_plain / n_pass * 1e3, 4)
            out["value_note"] = ("`value` includes the RCCL re-assembly of every pass (overlapped with the next pass's encode); "
                                 "`value_no_reassemble` is the same K steps with the exchange switched off, timed right after")
        if ra_path is not None:
            out["reassembly_path"] = ra_path
        if reasm_ms is not None:
            out["reassembly_ms"] = round(reasm_ms, 4)
            out["reassembly_note"] = ("device time per encode pass of the exchange (length all-gathers + grouped send/recv of the shard "
                                      "bytes), HIP events on the side stream it runs on, max over ranks; it overlaps the next encode")
        if sampled is not None:
            out["bit_exact_sampled"] = sampled
        if placement is not None:
            out["reassembly_check"] = placement
        if world == 1 and not args.no_decode:
            try:
                out["decode"] = decode_leg(torch, ctx, fmt, B, last["out"], last["offsets"], d_pcm, reps=max(3, min(args.steps, 10)))
            except Exception as e:
                out["decode"] = {"error": repr(e)}
        if world == 1 and args.cpu_packets > 0:
            n = min(args.cpu_packets, B)
            pcm_host = alac_amd.synth_pcm(first_frame, n, fmt)  # host generator: what the oracle encodes
            n_bytes = int(last["offsets"][n].item())
            g_stream = last["out"][:n_bytes].cpu().numpy()
            g_sizes = last["sizes"][:n].cpu().numpy().astype(np.uint32)
            base, exact, ref_stages = cpu_baseline(fmt, pcm_host, n, g_stream, g_sizes)
            out["cpu_baseline"] = base
            if ref_stages is not None:
                out["cpu_reference_stages"] = ref_stages
            out["bit_exact_vs_cpu"] = exact
            out["bit_exact_packets"] = n
            # against the FASTER single-thread CPU figure of this run: the reference's own compiled stages when they travelled
            denom = max(base["value"], ref_stages["value"] if ref_stages else 0.0)
            out["speedup_vs_cpu_1thread"] = round(value / denom, 1)
            out["speedup_vs_cpu_1thread_denominator"] = ("cpu_reference_stages" if ref_stages and ref_stages["value"] >= base["value"]
                                                         else "cpu_baseline")
            out["speedup_vs_cpu_port_1thread"] = round(value / base["value"], 1)
            out["cpu_all_cores"] = cpu_all_cores(fmt)
        if world == 1 and not args.no_legs and args.packets == 0 and args.bit_depth == 16:
            del bufs, last
            torch.cuda.empty_cache()
            for name, depth, packets, every in (("bits24_10k", 24, 10000, 97), ("shard_125k", 16, 125000, 997)):
                try:
                    out[name] = encode_leg(torch, alac_amd, ctx, alac_amd.make_format(4096, depth, 2, 44100), packets, every=every)
                except Exception as e:  # a leg never takes the headline down
                    out[name] = {"error": repr(e)}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if os.environ.get("ALAC_BENCH_STUB") and "WORLD_SIZE" in os.environ:
        return stub_rank_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # decided before torch / alac_amd are imported or any GPU call is made: the ranks are CHILD processes
        return self_launch(args, sys.argv[1:])
    return rank_main(args)


if __name__ == "__main__":
    sys.exit(main())
