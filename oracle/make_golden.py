#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

TEST INFRASTRUCTURE ONLY.  This script runs the reference's own numpy golden
generator, /root/reference/pipeline/verification/make_golden_inputs.py, in
this container (it never travels to the GPU box: /root/reference does not
exist there) and stores what it writes as small data fixtures:

  tests/golden/in_8t_4c_16s_2p_deadbeef.dat      (cfg 1, random, seed 0xdeadbeef)
  tests/golden/corr_8t_4a_4c_16s_2p_deadbeef.dat
  tests/golden/in_8t_4c_16s_2p_chanramp.dat      (cfg 1, --chanramp)
  tests/golden/corr_8t_4a_4c_16s_2p_chanramp.dat
  tests/golden/golden_64t_32a_8c_32s_2p_deadbeef.npz  (64-input case, compressed:
        'vin' uint8[T,C,S,P] and 'corr_re'/'corr_im' int32[T/acc,C,S,S,P,P])

File format of the .dat files is the reference's: one JSON header line then raw
bytes (uint8 input / complex128 visibilities [T/acc, chan, s0, s1, p0, p1] with
corr = x[s0,p0] * conj(x[s1,p1]), make_golden_inputs.py:156-158).

The only modification is the shim `np.complex = complex` (numpy >= 1.24 removed
the alias the script uses); no reference source is copied.
"""
import json
import os
import runpy
import sys
import tempfile

import numpy as np

REF = "/root/reference/pipeline/verification/make_golden_inputs.py"
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def run_ref(args, outdir):
    np.complex = complex  # shim for numpy >= 1.24
    argv = sys.argv
    sys.argv = [REF] + args + ["--datapath", outdir]
    try:
        try:
            runpy.run_path(REF, run_name="__main__")
        except SystemExit:
            pass
    finally:
        sys.argv = argv


def strip_time(path):
    """The reference stamps time.time() into the JSON header: zero it so the
    fixture is reproducible byte-for-byte."""
    with open(path, "rb") as fh:
        hdr = fh.readline()
        body = fh.read()
    meta = json.loads(hdr.decode())
    meta["time"] = 0
    with open(path, "wb") as fh:
        fh.write(json.dumps(meta).encode() + b"\n")
        fh.write(body)


def main():
    os.makedirs(GOLD, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        # cfg 1 (BASELINE.json configs[0]): 16 stands x 2 pol, 4 chan
        run_ref(["-t", "8", "-c", "4", "-s", "16", "-p", "2", "--accshort", "4"], td)
        run_ref(["-t", "8", "-c", "4", "-s", "16", "-p", "2", "--accshort", "4", "--chanramp"], td)
        for f in sorted(os.listdir(td)):
            strip_time(os.path.join(td, f))
            os.replace(os.path.join(td, f), os.path.join(GOLD, f))
    with tempfile.TemporaryDirectory() as td:
        # 64-input case (one full 64-input tile of the HIP kernel), compressed
        run_ref(["-t", "64", "-c", "8", "-s", "32", "-p", "2", "--accshort", "32"], td)
        fin = os.path.join(td, "in_64t_8c_32s_2p_deadbeef.dat")
        fco = os.path.join(td, "corr_64t_32a_8c_32s_2p_deadbeef.dat")
        with open(fin, "rb") as fh:
            fh.readline()
            vin = np.frombuffer(fh.read(), dtype=np.uint8).reshape(64, 8, 32, 2)
        with open(fco, "rb") as fh:
            fh.readline()
            corr = np.frombuffer(fh.read(), dtype=np.complex128).reshape(2, 8, 32, 32, 2, 2)
        assert np.all(corr.real == np.round(corr.real)) and np.all(corr.imag == np.round(corr.imag))
        np.savez_compressed(os.path.join(GOLD, "golden_64t_32a_8c_32s_2p_deadbeef.npz"),
                            vin=vin, corr_re=corr.real.astype(np.int32), corr_im=corr.imag.astype(np.int32))
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()
