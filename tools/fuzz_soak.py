"""One-off soak of tests/test_gpu_fuzz.py's randomised parity case with seeds beyond the 48 the suite runs:
    python tools/fuzz_soak.py FIRST LAST ["k=v,k=v"] [foreign]
                                    (GPU box; every seed: GPU packets == oracle packets, decode == input; the optional third
                                     argument pins context options, e.g. "thru=1"; "foreign": the randomised FOREIGN-stream
                                     decode case of tests/test_gpu_foreign.py instead — forged headers and cookies)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import alac_amd  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
import test_gpu_fuzz as fz  # noqa: E402


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    ctx, oracle = alac_amd.Context(0), Oracle()
    if len(sys.argv) > 3:
        for kv in sys.argv[3].split(","):
            if kv:
                ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        print("options", sys.argv[3], flush=True)
    case = fz.test_random_layouts_match_oracle_and_round_trip
    if len(sys.argv) > 4 and sys.argv[4] == "foreign":
        import test_gpu_foreign as ff
        case = ff.test_random_foreign_streams
        print("case: foreign streams", flush=True)
    case = getattr(case, "__wrapped__", case)
    bad = 0
    for seed in range(first, last):
        try:
            case(ctx, oracle, seed)
        except AssertionError as e:
            bad += 1
            print("seed", seed, "FAILED", str(e)[:200], flush=True)
        if (seed - first + 1) % 100 == 0:
            print("seeds", first, "..", seed, "failures", bad, flush=True)
    print("RESULT", "ok" if bad == 0 else "FAILED", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
