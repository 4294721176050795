"""Config 2 (64-QAM over flat Jakes fading, 1e5 symbols) rate against realizations per launch, both arithmetics and both
demodulators (the bench leg printed 4.68e6/s at 16 384 per launch, the profile 3.39e6/s at 65 536).
Runs on the GPU box: python scripts/c2_batch_sweep.py [out.json]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from pyphysim_amd.engine import Engine  # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "c2_batch_sweep.json")
    eng = Engine(0, "f64")
    rows = []
    for dt in ("f32", "f64"):
        for demod in ("slicer", "mindist"):
            run, units, wl = bench.make_runner(eng, "c2", demod, dt)
            cnt = eng.new_counters()
            run(1 << 40, 8192, cnt)
            eng.sync()
            for lg in range(13, 21):
                nb = 1 << lg
                if dt == "f64" and nb > (1 << 18):
                    continue
                reps = 3 if nb <= (1 << 17) else 1
                run((1 << 41) + nb, nb, cnt)           # warm-up at this size (scratch growth)
                eng.sync()
                eng.timer_start()
                for r in range(reps):
                    run((1 << 42) + r * nb, nb, cnt)
                ms = eng.timer_stop_ms() / reps
                rows.append({"dtype": dt, "demod": demod, "realizations_per_launch": nb, "ms_per_launch": ms,
                             "realizations_per_s": nb / ms * 1e3})
                print(rows[-1], flush=True)
    json.dump({"workload": wl, "rows": rows, "device": eng.device_name}, open(out_path, "w"), indent=1)
    eng.close()


if __name__ == "__main__":
    main()
