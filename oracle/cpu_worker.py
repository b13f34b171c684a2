"""TEST INFRASTRUCTURE (bench.py's cpu_all_cores leg): encode one synthetic 4096x4096 frame with the CPU reference
build (oracle/_ref; falls back to our restatement) and print the seconds taken.  One process per host core, because the
reference library is not re-entrant.   python -m oracle.cpu_worker <seed> <w> <h> <stages> <filt> <segments>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    seed, w, h, stages, filt, segments = (int(x) for x in sys.argv[1:7])
    from icer_compression_amd import synth
    from oracle import binding
    impl = binding.Reference() if binding.have_reference() else binding.Oracle()
    frame = synth.gray_frame(w, h, seed, 1)
    t = time.perf_counter()
    rc, stream, _ = impl.compress([frame], stages, filt, segments, 2 * w * h)
    print(f"{rc} {len(stream)} {time.perf_counter() - t:.4f}", flush=True)


if __name__ == "__main__":
    main()
