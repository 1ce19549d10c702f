"""Randomised soak of the verifier's paired terms (ZKP_OPT_JOINT_LADDER, round 6): the random-statement test of tests/test_gpu_fused.py -- randomly shaped statements
through the host route, the fused route and the oracle: proofs byte for byte, verdicts of valid and tampered proofs -- for as many seeds as SECONDS allow, under
five settings: pairs + tables of multiples on the throughput schedule, pairs on the latency schedule, pairs without the tables, separate terms on both schedules.

    python tools/soak_pairs.py [SECONDS = 240]        (GPU box)"""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from zkp_amd.engine import Engine
    from zkp_amd import toolbox as T
    import tests.test_gpu_fused as F
    t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 240.0)
    done, fails, seed = {}, [], 14
    while time.time() < t_end:
        for sched, opt in ((1, 1), (0, 1), (1, 2), (1, 0), (0, 0)):
            e = Engine(0)
            e.set_option(14, sched)       # ZKP_OPT_SYNC_SCHEDULE = 1: synchronous calls on the throughput schedule, where the riders' tables are built at any size
            e.set_option(17, opt)         # ZKP_OPT_JOINT_LADDER
            try:
                F.test_random_statements_fused_equals_host_route_and_oracle(e, seed)
            except AssertionError:
                tb = traceback.extract_tb(sys.exc_info()[2])[-1]
                fails.append((seed, sched, opt, tb.lineno))
                print("FAIL seed", seed, "ZKP_OPT_SYNC_SCHEDULE", sched, "ZKP_OPT_JOINT_LADDER", opt, "line", tb.lineno, tb.line, flush=True)
            finally:
                e.close()
                T.set_fused_min_batch(32)
            done[(sched, opt)] = done.get((sched, opt), 0) + 1
        seed += 1
    print("pairs soak: seeds 14 .. %d x %d settings (runs per (schedule, option): %s); failures: %d" % (seed - 1, len(done), done, len(fails)))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
